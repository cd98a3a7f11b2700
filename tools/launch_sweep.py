#!/usr/bin/env python3
"""launch_sweep.py [WORKLOAD]: kernel time of one launch against the number of frames it holds (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
from raytracing_amd import api, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "chair_4k"
scene, w, h = WORKLOADS[wl]
nodes = scenes.csg1024() if scene == "csg1024" else api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
info = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)[0:1]
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_batch(32)
    for n in (1, 2, 4, 8, 10, 12, 16, 20, 24, 32):
        rm.set_info_batch(np.repeat(info, n))
        for _ in range(4):
            rm.process()
        t = sorted(rm.process() for _ in range(11))[5]
        print("%s: %2d frames per launch: %.3f ms per launch, %.4f ms per frame" % (wl, n, t, t / n), flush=True)
