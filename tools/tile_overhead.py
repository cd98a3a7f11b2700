#!/usr/bin/env python3
"""tile_overhead.py: what handing out, setting up and storing a tile costs (GPU box).  Two scenes without a march:
a unit sphere a million units away (too far out for a ray window: every ray is marched, and leaves after one
step), and chair.sce seen from far below its lattice (every ray is dropped by its window: no step at all)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import api, scenes

info0 = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)[0:1]
below = info0.copy()
below["StartPos"] = (0.0, 0.0, -2500.0)
cases = [("one step per ray", scenes.Tree(("movo", (1e6, 0, 0), ("sfero", (1,)))).table(), info0),
         ("no step (all rays outside the window)", api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", "chair.sce")), below)]
# tile_overhead.py CASE W H BATCH: that one case only (under rocprofv3 --pmc: instructions per tile)
sizes, batches = ((3840, 2160), (7680, 4320)), (1, 16)
if len(sys.argv) > 4:
    cases = [cases[int(sys.argv[1])]]
    sizes, batches = ((int(sys.argv[2]), int(sys.argv[3])),), (int(sys.argv[4]),)
for label, table, info in cases:
    for w, h in sizes:
        for batch in batches:
            with api.Raymarching(w, h) as rm:
                rm.set_nodes(table)
                rm.set_batch(batch)
                rm.set_info(info)
                for _ in range(5):
                    rm.process()
                t = sorted(rm.process() for _ in range(21))[10]
                tiles = (w // 8) * (h // 8) * batch
                print("%s: %dx%d, %d frame(s) per launch: %.4f ms per frame, %.2f ns per tile over the chip" %
                      (label, w, h, batch, t / batch, t * 1e6 / tiles), flush=True)
