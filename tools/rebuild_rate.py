#!/usr/bin/env python3
"""Rank 0's extra work at N>1, alone on the GPU: rm_assemble_luminance_batch for a launch of 32 frames whose
bands 8 ranks hold (GPU box).  Prints microseconds per frame and the HBM rate (5 B/px: 1 read, 4 written)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import api, tiling

sizes = ((3840, 2160), (7680, 4320)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),)     # rebuild_rate.py [W H]
for w, h in sizes:
    nranks, band, n = 8, 16, 32
    for weights in ([1] * 8, [3] + [5] * 7):
        owner = tiling.deal(weights)
        rows = -(-tiling.max_local_rows(h, band, nranks, owner) // 16) * 16
        parts = torch.randint(0, 255, (nranks, n, rows, w), dtype=torch.uint8, device="cuda")
        out = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        with api.Raymarching(w, h) as rm:
            rm.set_bands(band, 0, nranks, owner)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(6):
                if k == 1:
                    e0.record()
                rm.assemble_luminance_batch(parts.data_ptr(), n * rows * w, rows * w, w, n, out.data_ptr(), w * 4, h * w * 4,
                                            stream=torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * n)
            print("%dx%d weights %s: %.2f us per frame, %.0f GB/s" % (w, h, weights, us, 5 * w * h / us / 1e3), flush=True)
        del parts, out
