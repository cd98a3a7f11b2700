#!/usr/bin/env python3
"""Renders a workload a few times (profiling target): one_frame.py SCENE W H [REPS] [norm_batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_info(O.default_info(1, 0))
    if len(sys.argv) > 5:
        rm.set_norm_batch(int(sys.argv[5]))
    print([round(rm.process(), 4) for _ in range(reps)], rm.scene_stats())
