#!/usr/bin/env python3
"""slice_rate.py NRANKS [scene W H]: what one GPU of an N-rank job sustains, without the gather:
contexts rendering rank 0's bands of the frame, k of them taking frames in turn on streams of
their own, for k = 1..8.  Prints band-sets per second next to N x the whole-frame rate (GPU box).
Environment: GPU_MAX_HW_QUEUES as given (HIP's default is 4)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle as O          # camera block only
from raytracing_amd import api, tiling

nranks = int(sys.argv[1])
scene = sys.argv[2] if len(sys.argv) > 2 else "chair"
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
info = O.default_info(1, 0)
band = tiling.default_band_rows(h, nranks) if nranks > 1 else 0
batch = int(os.environ.get("SLICE_BATCH", "1"))        # frames per launch (rm_set_batch)


def rate(k, frames=600 // batch):
    ctxs, streams = [], []
    for _ in range(k):
        rm = api.Raymarching(w, h)
        rm.set_nodes(nodes)
        rm.set_info(info)
        if band:
            rm.set_bands(band, 0, nranks)
        if batch > 1:
            rm.set_batch(batch)
        st = torch.cuda.Stream()
        rm.set_stream(st.cuda_stream)
        ctxs.append(rm)
        streams.append(st)
    for i in range(4 * k):
        ctxs[i % k].process_async()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        ctxs[i % k].process_async()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    for rm in ctxs:
        rm.close()
    return frames * batch / secs, t_host / frames * 1e6


whole = None
print("GPU_MAX_HW_QUEUES=%s  %s %dx%d  %d ranks, bands of %d, %d frame(s) per launch" %
      (os.environ.get("GPU_MAX_HW_QUEUES", "default"), scene, w, h, nranks, band, batch))
for k in (1, 2, 3, 4, 6, 8):
    fps, host_us = rate(k)
    print("  %d in flight: %8.0f band-sets/s  (%.1f us of host time per launch)" % (k, fps, host_us), flush=True)
