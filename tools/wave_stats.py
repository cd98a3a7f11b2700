#!/usr/bin/env python3
"""Per-wave diagnostics of a frame: wave_stats.py SCENE W H"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if scene == "csg1024":
    from raytracing_amd import scenes
    nodes = scenes.csg1024()
else:
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
if os.environ.get("RM_JIT"): print("RM_JIT", os.environ["RM_JIT"])
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_batch(int(os.environ.get("BATCH", "1")))
    rm.set_info(O.default_info(1, 0))
    for _ in range(int(os.environ.get("WARM_FRAMES", "0"))):      # let the tile order settle first
        rm.process()
    ms = rm.process()
    st = rm.wave_stats().astype(np.float64)
ev, nm, cyc, ncyc = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
print("kernel %.3f ms, waves %d" % (ms, len(st)))
for name, v in (("trips", ev), ("norm passes", nm), ("cycles", cyc), ("norm cycles", ncyc)):
    print("%-12s mean %10.1f  p50 %8.0f  p90 %8.0f  p99 %8.0f  max %8.0f  sum %.3e" %
          (name, v.mean(), *np.percentile(v, [50, 90, 99]), v.max(), v.sum()))
busy = nm > 0
if busy.any():
    print("waves with normals: %d; cycles per norm pass %.0f; cycles per trip excluding norms %.0f" %
          (busy.sum(), ncyc[busy].sum() / nm[busy].sum(), (cyc - ncyc).sum() / ev.sum()))
i = int(np.argmax(cyc))
print("slowest wave: trips %d, norm passes %d, cycles %d, norm cycles %d" % tuple(st[i]))
