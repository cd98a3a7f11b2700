import sys, time
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
from raytracing_amd import api, scenes
for leaves in (341, 682):
    t = scenes.csg_union(17, leaves)
    with api.Raymarching(3840, 2160) as rm:
        t0 = time.time(); rm.set_nodes(t); dt = time.time() - t0
        print(leaves, "leaves:", rm.scene_stats(), "set_nodes %.1f s" % dt, flush=True)
        rm.set_info(O.default_info(1, 0))
        for _ in range(3): rm.process()
        print("  4K frame %.2f ms" % min(rm.process() for _ in range(5)), flush=True)
