#!/usr/bin/env python3
"""Diagnostic: NaN cells of the value+grad sweep of cfg2's scene with / without the last-segment masks and in the exhaustive kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_scene
from differt2d_amd.engine import Context
F = np.float32
tx, walls = random_scene(50, seed=1234)
x = np.linspace(0.0, 1.0, 1024).astype(F)
X, Y = np.meshgrid(x, x)
for kw in (dict(approx=False), dict(approx=True, function="hard_sigmoid")):
    res = {}
    for name, opts, strict in (("on", {}, False), ("off", {"hidden_masks": 0}, False), ("strict", {}, True)):
        with Context(0) as c:
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_scene(walls)
            for _ in range(3):
                g = c.value_and_grads(tx, X, Y, min_order=0, max_order=2, strict_nan=strict, **kw)
            res[name] = np.isnan(g["grad_rx"]).any(-1)
    on, off, st = res["on"], res["off"], res["strict"]
    print(kw, "NaN cells: on", int(on.sum()), "off", int(off.sum()), "strict", int(st.sum()),
          "| on&~strict", int((on & ~st).sum()), "strict&~on", int((st & ~on).sum()), "off&~strict", int((off & ~st).sum()), "strict&~off", int((st & ~off).sum()))
    print("   cells strict&~on:", np.argwhere(st & ~on)[:10].tolist(), " on&~off:", np.argwhere(on & ~off)[:10].tolist(), " off&~on:", np.argwhere(off & ~on)[:10].tolist())
