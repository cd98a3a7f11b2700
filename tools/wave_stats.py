#!/usr/bin/env python3
"""Per-wave timeline of one launch: wave_stats.py SCENE W H   (env BATCH, WARM_FRAMES)
Where a launch's time goes: when the wavefronts start and leave, how evenly the work is spread, what the
longest tile costs.  s_memtime ticks at 100 MHz: 10 ns each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import api

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if scene == "csg1024":
    from raytracing_amd import scenes
    nodes = scenes.csg1024()
else:
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
info = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)[0:1]
batch = int(os.environ.get("BATCH", "1"))
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_batch(batch)
    rm.set_info(info)
    for _ in range(int(os.environ.get("WARM_FRAMES", "3"))):      # let the tile order settle first
        rm.process()
    ms = sorted(rm.process() for _ in range(9))[4]
    import time
    t_host = time.perf_counter()
    st = rm.wave_stats().astype(np.int64)
    t_host = time.perf_counter() - t_host
out = os.path.join(ROOT, "gpurun_out", "ws_raw_%s_%dx%d_b%d.npy" % (scene, w, h, batch))
np.save(out, st)
print("stats launch took %.3f ms on the host (allocation and copies included); raw array in %s" % (t_host * 1e3, out))
ev, nm, cyc, ncyc, t0, t1, tiles, longest = (st[:, k] for k in range(8))
# start and end are readings of the 100 MHz clock (s_memrealtime), shared by the XCDs: 10 ns per tick
xcd = (np.arange(len(st)) // 4) % 8
base = t0.min()
start = ((t0 - base) % (1 << 32)) * 0.01
end = ((t1 - base) % (1 << 32)) * 0.01
print("%s %dx%d batch %d: kernel %.3f ms (median of 9 timed launches); the instrumented launch spans %.1f us; waves %d" %
      (scene, w, h, batch, ms, end.max(), len(st)))
print("wave start  us: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile(start, [50, 90, 99, 100])))
print("wave end    us: p1 %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(end, [1, 10, 50, 90, 100])))
print("wave-time in the kernel / (span x waves) = %.3f" % ((end - start).sum() / (end.max() * len(st))))
print("trips: total %.3e, per wave mean %.0f max %d; tiles per wave mean %.1f max %d; longest tile %d trips" %
      (ev.sum(), ev.mean(), ev.max(), tiles.mean(), tiles.max(), longest.max()))
dur = end - start
ok = ev > 0
i = int(np.argmax(longest))
print("us per trip, mean over waves: %.4f; of the wave with the longest tile: %.4f" % (dur[ok].sum() / ev[ok].sum(), dur[i] / max(1, ev[i])))
print("wave with the longest tile: start %.1f end %.1f us, trips %d, tiles %d, longest %d" % (start[i], end[i], ev[i], tiles[i], longest[i]))
edges = np.linspace(0, end.max(), 21)
alive = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5% slice of the span:", alive)
# trips in flight per slice (a wave's trips spread evenly over its life): where the chip is short of work
work = np.zeros(20)
for a_, b_, n_ in zip(start, end, ev):
    if b_ <= a_:
        continue
    lo, hi = np.clip((a_ - edges[:-1]) / (edges[1] - edges[0]), 0, 1), np.clip((b_ - edges[:-1]) / (edges[1] - edges[0]), 0, 1)
    work += (hi - lo) * n_ / ((b_ - a_) / (edges[1] - edges[0]))
print("trips per 5% slice (k):", [int(v / 1e3) for v in work])
for j in np.argsort(end)[-5:]:
    print("  late wave: xcd %d start %.1f end %.1f trips %d tiles %d longest %d" % (xcd[j], start[j], end[j], ev[j], tiles[j], longest[j]))
