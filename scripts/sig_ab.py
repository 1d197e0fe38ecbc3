#!/usr/bin/env python3
"""cfg2 in sigmoid validity: options A/B (same bits expected), kernel time and the map's CRC.
usage: sig_ab.py option=v1,v2,... [option=...]"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

tx, walls, X, Y = workload(50, 1024)
opts = [a.split("=") for a in sys.argv[1:]]
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    ctx.set_option("time_kernel", 1)
    p = make_params(min_order=0, max_order=2, approx=True, function="sigmoid")
    for name, vals in [(None, [None])] + [(k, v.split(",")) for k, v in opts]:
        for v in vals:
            if name:
                ctx.set_option(name, int(v))
            for _ in range(3):
                ctx.launch(p, tx)
            km = []
            for _ in range(8):
                ctx.launch(p, tx)
                km.append(ctx.last_kernel_ms())
            Z = ctx.get_map()
            print(f"{name}={v}: kernel {np.mean(km):.3f} ms (min {np.min(km):.3f}), crc {zlib.crc32(Z.tobytes()):08x}", flush=True)
