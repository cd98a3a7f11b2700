#!/usr/bin/env python3
"""frame_series.py SCENE W H [N]: kernel ms of N single-frame launches in a row (tile order fed forward) - is the
time steady, or does it alternate?  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import api

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
info = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)[0:1]
print("RM_JIT_FLAGS =", os.environ.get("RM_JIT_FLAGS", ""))
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_info(info)
    ms = [rm.process() for _ in range(n)]
    print("sync each:", " ".join("%.3f" % v for v in ms))
    # back to back without waiting in between: wall time per frame
    import time
    rm.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        rm.process_async()
    rm.synchronize()
    print("back to back: %.4f ms per frame" % ((time.perf_counter() - t0) * 1e3 / n))
    rm.set_heavy_first(False)
    ms = [rm.process() for _ in range(10)]
    print("natural order:", " ".join("%.3f" % v for v in ms))
