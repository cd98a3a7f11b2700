#!/usr/bin/env python3
"""single_frame_probe.py: median and minimum kernel time per frame, one frame per launch and eight, stock
scenes at three sizes and csg1024; run under different RM_JIT_FLAGS (e.g. -DRM_SINGLE_CHUNK=1) to compare builds (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import api
INFO = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)[0:1]

print("RM_JIT_FLAGS =", os.environ.get("RM_JIT_FLAGS", ""))
from raytracing_amd import scenes
CASES = (("chair", 3840, 2160), ("chair", 1920, 1080), ("chair", 7680, 4320), ("infinity", 3840, 2160), ("csg1024", 3840, 2160))
if os.environ.get("PROBE_FAST"):
    CASES = CASES[:4]
for scene, w, h in CASES:
    nodes = scenes.csg1024() if scene == "csg1024" else api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    reps = 9 if scene == "csg1024" else 41
    for batch in (1, 8):
        with api.Raymarching(w, h) as rm:
            rm.set_nodes(nodes)
            rm.set_batch(batch)
            rm.set_info(INFO)
            for _ in range(4 if scene == "csg1024" else 8):
                rm.process()
            t = sorted(rm.process() / batch for _ in range(reps))
            print("  %-8s %dx%d  %d per launch: median %.4f ms  min %.4f ms per frame" % (scene, w, h, batch, t[reps // 2], t[0]), flush=True)
