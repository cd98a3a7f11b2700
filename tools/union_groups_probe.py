#!/usr/bin/env python3
"""union_groups_probe.py: csg1024 at 4K against the number of groups its union is cut into
(RM_UNION_GROUPS, read when the scene is set), 8 frames per launch and one (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api, scenes

nodes = scenes.csg1024()
for groups in [int(v) for v in sys.argv[1:]] or [6, 12, 20, 32]:
    os.environ["RM_UNION_GROUPS"] = str(groups)
    for batch in (8, 1):
        with api.Raymarching(3840, 2160) as rm:
            rm.set_nodes(nodes)
            rm.set_batch(batch)
            rm.set_info(O.default_info(1, 0))
            for _ in range(3):
                rm.process()
            t = sorted(rm.process() for _ in range(7))
            print("%d groups, %d per launch: median %.3f ms per frame, %d VGPRs" %
                  (groups, batch, t[3] / batch, rm.scene_stats()["kernel_vgprs"]), flush=True)
