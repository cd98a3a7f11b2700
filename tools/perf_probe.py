#!/usr/bin/env python3
"""Times the render kernel over a few workloads and knob settings (GPU box):
perf_probe.py [scene ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api, scenes as S


def time_it(nodes, w, h, info, reps=7, jit=1, **knobs):
    with api.Raymarching(w, h) as rm:
        rm.set_jit(jit)
        rm.set_nodes(nodes)
        rm.set_info(info)
        if "evaluator" in knobs:
            rm.set_evaluator(knobs["evaluator"])
        if "heavy" in knobs:
            rm.set_heavy_first(bool(knobs["heavy"]))
        if "boost" in knobs:
            rm.set_boost_age(knobs["boost"])
        rm.process()
        ms = sorted(rm.process() for _ in range(reps))
        return ms[len(ms) // 2]


def main():
    info = O.default_info(1, 0)
    G = os.path.join(ROOT, "tests", "golden", "scenes")
    scenes = {n: api.load_sce(os.path.join(G, n + ".sce")) for n in ("chair", "infinity")}
    scenes["csg1024"] = S.csg1024()
    jobs = [("chair", 3840, 2160), ("chair", 1920, 1080), ("chair", 7680, 4320), ("infinity", 3840, 2160), ("csg1024", 960, 540)]
    only = sys.argv[1:]
    for name, w, h in jobs:
        if only and name not in only:
            continue
        base = time_it(scenes[name], w, h, info)
        line = "%-9s %4dx%-4d  default %8.3f ms  %8.1f Mrays/s |" % (name, w, h, base, w * h / base / 1e3)
        for ba in (8, 24, 96, 0xffffffff):
            line += " ba%d %.3f" % (ba if ba < 1000 else -1, time_it(scenes[name], w, h, info, boost=ba))
        line += " | heavy0 %.3f heavy0,ba-1 %.3f |" % (time_it(scenes[name], w, h, info, heavy=0),
                                                      time_it(scenes[name], w, h, info, heavy=0, boost=0xffffffff))
        if name != "csg1024":
            line += " | program %.3f" % time_it(scenes[name], w, h, info, reps=3, jit=0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
