import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import workload
from differt2d_amd.engine import Context, make_params
tx, walls, X, Y = workload(50, 1024)
names = ["cand_eval","reached_loss","reached_occl","nonzero_valid","seg_tests","exact_div","sum_k","sum_k_loss","sum_k1_nz","cull_levels","t_prologue","t_o0","t_o1","t_o2","t_exact","x"]
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for mode, kw in (("hard", {}), ("hsig", dict(approx=True)), ("sig", dict(approx=True, function="sigmoid"))):
        p = make_params(min_order=0, max_order=2, **kw)
        for _ in range(3): ctx.launch(p, tx)
        ctx.synchronize()
        st = ctx.launch_stats(p, tx)
        tiles = 128*128
        print(mode, {n: round(float(v)/tiles, 2) for n, v in zip(names, st)})
        Z = ctx.get_map()
        print("   zero cells", float((Z == 0).mean()), "tiny (<1e-30) cells", float((np.abs(Z) < 1e-30).mean()))
