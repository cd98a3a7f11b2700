#!/usr/bin/env python3
"""chunk_probe.py: chair 4K, one frame per launch and 8 per launch, median kernel time per frame (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

scene = sys.argv[1] if len(sys.argv) > 1 else "chair"
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
w, h = 3840, 2160
for batch in (1, 8):
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(nodes)
        rm.set_batch(batch)
        rm.set_info(O.default_info(1, 0))
        for _ in range(8):
            rm.process()
        t = sorted(rm.process() for _ in range(31))
        print("%s flags %r: %d per launch: median %.4f ms per frame, min %.4f, max %.4f" %
              (scene, os.environ.get("RM_JIT_FLAGS", ""), batch, t[15] / batch, t[0] / batch, t[-1] / batch), flush=True)
