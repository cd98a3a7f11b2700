#!/usr/bin/env python3
"""ab_knob.py KNOB [scene W H]: median kernel time with a library switch on and off, interleaved
(GPU box).  KNOB: culling | short_math | heavy_first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

knob = sys.argv[1]
scene = sys.argv[2] if len(sys.argv) > 2 else "chair"
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
ctx = {}
for on in (0, 1):
    rm = api.Raymarching(w, h)
    getattr(rm, "set_" + knob)(bool(on))
    rm.set_nodes(nodes)
    rm.set_info(O.default_info(1, 0))
    for _ in range(5):
        rm.process()
    ctx[on] = rm
times = {0: [], 1: []}
for rep in range(40):
    for on in (0, 1):
        times[on].append(ctx[on].process())
for on in (0, 1):
    t = sorted(times[on])
    print("%s=%d  %s %dx%d  median %.4f ms  min %.4f ms" % (knob, on, scene, w, h, t[len(t) // 2], t[0]), flush=True)
