#!/usr/bin/env python3
"""overlap_probe.py SCENE W H: frames per second with 1, 2 and 3 contexts (streams) taking frames in turn."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from raytracing_amd import api

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
info = O.default_info(1, 0)
for nctx in (1, 2, 3, 4):
    ctxs = []
    for _ in range(nctx):
        rm = api.Raymarching(w, h)
        rm.set_nodes(nodes)
        rm.set_info(info)
        ctxs.append(rm)
    for k in range(20):
        ctxs[k % nctx].process_async()
    for rm in ctxs:
        rm.synchronize()
    n = 300
    t0 = time.perf_counter()
    for k in range(n):
        ctxs[k % nctx].process_async()
    for rm in ctxs:
        rm.synchronize()
    dt = time.perf_counter() - t0
    print("%s %dx%d, %d context(s): %.3f ms/frame, %.1f Mrays/s" % (scene, w, h, nctx, dt / n * 1e3, w * h * n / dt / 1e6), flush=True)
    for rm in ctxs:
        rm.close()
