#!/usr/bin/env python3
"""tile_overhead.py: a scene every ray leaves after one step (a unit sphere a million units away):
the frame time is then the cost of handing out, setting up and storing tiles (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api, scenes

table = scenes.Tree(("movo", (1e6, 0, 0), ("sfero", (1,)))).table()
for w, h in ((3840, 2160), (7680, 4320)):
    for batch in (1, 8):
        with api.Raymarching(w, h) as rm:
            rm.set_nodes(table)
            rm.set_batch(batch)
            rm.set_info(O.default_info(1, 0))
            for _ in range(5):
                rm.process()
            t = sorted(rm.process() for _ in range(21))[10]
            tiles = (w // 8) * (h // 8) * batch
            print("%dx%d, %d frame(s) per launch: %.4f ms per frame, %.2f ns per tile over the chip" %
                  (w, h, batch, t / batch, t * 1e6 / tiles), flush=True)
