#!/usr/bin/env python3
"""slice_rate.py NRANKS [scene W H]: what one GPU of an N-rank job sustains, without the gather:
contexts rendering rank 0's bands of the frame, k of them taking frames in turn on streams of
their own, for k = 1..8.  Prints band-sets per second next to N x the whole-frame rate (GPU box).
Environment: GPU_MAX_HW_QUEUES as given (HIP's default is 4); SLICE_BATCH frames per launch;
SLICE_WEIGHTS=w0,w1,... a weighted deal of the bands (tiling.deal) and SLICE_RANK the rank whose share is
rendered (default 0); SLICE_REBUILD=1: the rank also rebuilds every launch's RGBA8 frames from the planes
of all ranks on a second stream, as rank 0 of bench.py does (the planes of the other ranks are zeros: the
kernel's work does not depend on what they hold)."""
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
weights = [int(x) for x in os.environ["SLICE_WEIGHTS"].split(",")] if os.environ.get("SLICE_WEIGHTS") else [1] * nranks
owner = tiling.deal(weights) if nranks > 1 else None
rank = int(os.environ.get("SLICE_RANK", "0"))
rebuild = os.environ.get("SLICE_REBUILD", "0") == "1" and nranks > 1
rows_max = -(-tiling.max_local_rows(h, band, nranks, owner) // 16) * 16 if band else h


def rate(k, frames=600 // batch):
    ctxs, streams, slabs = [], [], []
    comm = torch.cuda.Stream()
    out = torch.zeros((batch, h, w, 4), dtype=torch.uint8, device="cuda") if rebuild else None
    done = [torch.cuda.Event() for _ in range(k)]
    free = [torch.cuda.Event() for _ in range(k)]

    def launch(i):
        j = i % k
        if rebuild:
            streams[j].wait_event(free[j])
        ctxs[j].process_async()
        if rebuild:
            done[j].record(streams[j])
            comm.wait_event(done[j])
            ctxs[j].assemble_luminance_batch(slabs[j].data_ptr(), batch * rows_max * w, rows_max * w, w, batch,
                                             out.data_ptr(), w * 4, h * w * 4, stream=comm.cuda_stream)
            free[j].record(comm)

    for _ in range(k):
        rm = api.Raymarching(w, h)
        rm.set_nodes(nodes)
        rm.set_info(info)
        if band:
            rm.set_bands(band, rank, nranks, owner)
        if rebuild:
            slabs.append(torch.zeros((nranks, batch, rows_max, w), dtype=torch.uint8, device="cuda"))
            rm.set_luminance_output(slabs[-1][rank].data_ptr(), w)
            rm.set_batch(batch, 0, rows_max * w)
        elif batch > 1:
            rm.set_batch(batch)
        st = torch.cuda.Stream()
        rm.set_stream(st.cuda_stream)
        ctxs.append(rm)
        streams.append(st)
    for i in range(4 * k):
        launch(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        launch(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    for rm in ctxs:
        rm.close()
    return frames * batch / secs, t_host / frames * 1e6


whole = None
print("GPU_MAX_HW_QUEUES=%s  %s %dx%d  %d ranks, bands of %d dealt %s, rank %d%s, %d frame(s) per launch" %
      (os.environ.get("GPU_MAX_HW_QUEUES", "default"), scene, w, h, nranks, band, weights, rank,
       " + rebuild of every frame" if rebuild else "", batch))
for k in [int(x) for x in os.environ.get("SLICE_K", "1,2,3,4,6,8").split(",")]:
    fps, host_us = rate(k)
    print("  %d in flight: %8.0f band-sets/s  (%.1f us of host time per launch)" % (k, fps, host_us), flush=True)
