#!/usr/bin/env python3
"""single_frame_probe.py: median and minimum kernel time of one frame per launch, stock scenes at three
sizes; run under different RM_JIT_FLAGS (e.g. -DRM_SINGLE_CHUNK=1) to compare builds (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

print("RM_JIT_FLAGS =", os.environ.get("RM_JIT_FLAGS", ""))
for scene, w, h in (("chair", 3840, 2160), ("chair", 1920, 1080), ("chair", 7680, 4320), ("infinity", 3840, 2160)):
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(nodes)
        rm.set_info(O.default_info(1, 0))
        for _ in range(8):
            rm.process()
        t = sorted(rm.process() for _ in range(41))
        print("  %-8s %dx%d  median %.4f ms  min %.4f ms" % (scene, w, h, t[20], t[0]), flush=True)
