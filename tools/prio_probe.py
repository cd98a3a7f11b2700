#!/usr/bin/env python3
"""prio_probe.py: per-frame kernel time against the steps of the long-tile priority (RM_PRIO_LEVELS and
RM_PRIO_STEP2/3 through RM_JIT_FLAGS, boost age through the API), all variants interleaved in one process (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

variants = [(48, 1, 3, 6), (48, 3, 3, 6), (48, 2, 3, 6), (24, 1, 3, 6), (96, 1, 3, 6), (48, 3, 1, 1)]      # the last: straight to priority 3, the first build
for scene, w, h, batch in (("chair", 3840, 2160, 8), ("infinity", 3840, 2160, 8), ("chair", 3840, 2160, 1), ("infinity", 3840, 2160, 1)):
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    ctxs = []
    for age, levels, s2, s3 in variants:
        os.environ["RM_JIT_FLAGS"] = "-DRM_PRIO_LEVELS=%d -DRM_PRIO_STEP2=%d -DRM_PRIO_STEP3=%d" % (levels, s2, s3)
        rm = api.Raymarching(w, h)
        rm.set_nodes(nodes)
        rm.set_batch(batch)
        rm.set_boost_age(age)
        rm.set_info(O.default_info(1, 0))
        for _ in range(6):
            rm.process()
        ctxs.append(rm)
    times = [[] for _ in variants]
    for rep in range(15):
        for i, rm in enumerate(ctxs):
            times[i].append(rm.process() / batch)
    for (age, levels, s2, s3), t in zip(variants, times):
        t.sort()
        print("%-8s %d per launch  age %3d, %d level(s) x%d x%-2d  median %.4f  min %.4f ms per frame" % (scene, batch, age, levels, s2, s3, t[len(t) // 2], t[0]), flush=True)
    for rm in ctxs:
        rm.close() if hasattr(rm, "close") else rm.__exit__(None, None, None)
