#!/usr/bin/env python3
"""perf_grid.py SCENE W H key=v1,v2 key=v1,v2 ...: kernel ms over the cross product of knob values.
keys: boost, heavy, evaluator, jit"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from perf_probe import time_it
from oracle import oracle as O
from raytracing_amd import api, scenes as S

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nodes = S.csg1024() if scene == "csg1024" else api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
keys, vals = [], []
for a in sys.argv[4:]:
    k, v = a.split("=")
    keys.append(k)
    vals.append([int(x) if int(x) >= 0 else 0xffffffff for x in v.split(",")])
info = O.default_info(1, 0)
for combo in itertools.product(*vals):
    kn = dict(zip(keys, combo))
    jit = kn.pop("jit", 1)
    ms = time_it(nodes, w, h, info, jit=jit, **kn)
    print("%s %dx%d %s -> %.3f ms" % (scene, w, h, " ".join("%s=%d" % (k, v if v < 1 << 31 else -1) for k, v in zip(keys, combo)), ms), flush=True)
