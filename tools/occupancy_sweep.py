#!/usr/bin/env python3
"""occupancy_sweep.py SCENE W H: kernel time and idle waves against workgroups per CU (GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 4:                      # child: one setting
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import oracle as O          # camera block only
    from raytracing_amd import api
    scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(nodes)
        rm.set_info(O.default_info(1, 0))
        for _ in range(5):
            rm.process()
        t = sorted(rm.process() for _ in range(31))
        st = rm.wave_stats()
        print("blocks/CU %s: median %.4f ms  min %.4f ms  waves %d, without a tile %d" %
              (sys.argv[4], t[15], t[0], len(st), int((st[:, 0] == 0).sum())), flush=True)
else:
    for k in (4, 5, 6, 7, 8):
        env = dict(os.environ, RM_BLOCKS_PER_CU=str(k))
        subprocess.run([sys.executable, __file__] + sys.argv[1:4] + [str(k)], env=env, check=True, timeout=120)
