#!/usr/bin/env python3
"""async_pageable_copy.py: does hipMemcpyAsync(device -> pageable host) + hipStreamSynchronize always
leave the bytes in place?  (One hit-mask miscompare in a suite run made the question worth a program;
rm_read_pixels now uses the blocking copy.)  Fresh numpy destinations of the sizes the tests read,
a memset to a new value before every copy.  GPU box."""
import ctypes as C
import sys

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
stream = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0          # hipStreamNonBlocking
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), 1 << 20) == 0
bad = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
keep = []
for i in range(rounds):
    size = (24000, 96000, 307200, 4096)[i & 3]
    v = 1 + i % 250
    assert hip.hipMemsetAsync(dev, v, size, stream) == 0
    out = np.zeros(size, np.uint8)
    assert hip.hipMemcpyAsync(C.c_void_p(out.ctypes.data), dev, size, 2, stream) == 0
    assert hip.hipStreamSynchronize(stream) == 0
    if not (out == v).all():
        bad += 1
        wrong = np.flatnonzero(out != v)
        print("round %d size %d: %d bytes wrong, first %d last %d, holds %s" % (i, size, len(wrong), wrong[0], wrong[-1], np.unique(out[wrong])[:4]), flush=True)
    keep.append(out)
    if len(keep) > 7:
        keep.pop(i % 7)                      # vary what the allocator hands back
print("%d rounds, %d bad" % (rounds, bad))
