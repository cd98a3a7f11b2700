#!/usr/bin/env python3
"""readback_rate.py: the frame rate when every frame also crosses to host memory through
rm_read_pixels (the PCIe-inclusive figure DESIGN.md quotes beside bench.py's HBM-resident one);
pageable numpy destination and a pinned one (GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # camera block only
from raytracing_amd import api

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", "chair.sce"))
torch.cuda.init()
with api.Raymarching(w, h) as rm:
    rm.set_nodes(nodes)
    rm.set_info(O.default_info(1, 0))
    pageable = np.zeros((h, w, 4), np.uint8)
    pinned = torch.zeros((h, w, 4), dtype=torch.uint8).pin_memory()
    for name, ptr in (("pageable", pageable.ctypes.data), ("pinned", pinned.data_ptr())):
        for _ in range(3):
            rm.process()
            api._check(api.lib().rm_read_pixels(rm._ctx, C.c_void_p(ptr), w * 4), "rm_read_pixels")
        n, kernel = 20, 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            kernel += rm.process()
            api._check(api.lib().rm_read_pixels(rm._ctx, C.c_void_p(ptr), w * 4), "rm_read_pixels")
        dt = (time.perf_counter() - t0) / n * 1e3
        print("%dx%d, %s host frame: %.3f ms per frame (kernel %.3f ms), %.0f Mrays/s, copy %.1f GB/s" %
              (w, h, name, dt, kernel / n, w * h / dt / 1e3, w * h * 4 / ((dt - kernel / n) * 1e6)), flush=True)
    got = pinned.numpy()
    assert np.array_equal(got, pageable)
