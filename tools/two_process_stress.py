#!/usr/bin/env python3
"""two_process_stress.py [PROCESSES [LAUNCHES]]: several processes rendering on ONE GPU at once (GPU box).

What the 2-rank rehearsal of bench.py does to the device, without bench.py around it: every process renders
chair.sce 1920x1080 in launches of 4 frames on three contexts in turn and checks completion every 12 launches.
Wavefronts of one process are preempted for the other's, which is what a persistent kernel must survive: in
round 3 the rehearsal met RM_ERR_INCOMPLETE twice in five runs.  Prints, per process, launches made, checks
failed, and for a failed check what it compared (tiles finished, tiles launched, watchdog flag).
RM_JIT_FLAGS=-DRM_WATCHDOG_SHADER_CLOCK=1 builds the kernels with the watchdog clock of rounds 1-2."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(launches):
    import numpy as np
    from raytracing_amd import api
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", "chair.sce"))
    info = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    ctxs = []
    for _ in range(3):
        rm = api.Raymarching(1920, 1080)
        rm.set_nodes(nodes)
        rm.set_batch(4)
        ctxs.append(rm)
    failed, seen = 0, []
    for k in range(launches):
        rm = ctxs[k % 3]
        rm.set_info_batch(info[k % 100:k % 100 + 4 - (k % 2)])       # launches of 4 and 3 frames in turn
        rm.process_async()
        if k % 12 == 11 or k == launches - 1:
            for c in ctxs:
                try:
                    c.synchronize()
                except api.RaymarchError as e:
                    if e.status != -10:
                        raise
                    failed += 1
                    seen.append(c.last_check())
    print("pid %d: %d launches, %d checks failed %s" % (os.getpid(), launches, failed, seen[:4]), flush=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2])))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    launches = sys.argv[2] if len(sys.argv) > 2 else "1200"
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", launches]) for _ in range(procs)]
    sys.exit(max(k.wait() for k in kids))
