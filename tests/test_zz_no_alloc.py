"""rm_process_static, the entry point that allocates nothing, against rm_process (GPU, through the C ABI)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from raytracing_amd import api

pytestmark = pytest.mark.gpu


def test_static_entry_point_matches_rm_process(stock_scenes):
    """Two calls in a row on one device, the second with a smaller image (so it finds the first call's
    counters and tile lists in the shared device array), then calls that alternate sizes and scenes: every
    frame byte-identical to rm_process's on the same context and to the oracle's."""
    torch = pytest.importorskip("torch")
    cases = [("chair", 640, 360, 0), ("chair", 200, 120, 3), ("chair", 200, 120, 3), ("infinity", 333, 190, 7),
             ("chair", 640, 360, 5), ("chair", 640, 360, 5)]
    ctxs = {}
    for name, w, h, phi in cases:
        if (name, w, h) not in ctxs:
            rm = api.Raymarching(w, h)
            rm.set_nodes(stock_scenes[name])
            ctxs[(name, w, h)] = rm
    try:
        for k, (name, w, h, phi) in enumerate(cases):
            rm = ctxs[(name, w, h)]
            info = O.default_info(1 + k, phi)
            pitch = w * 4 + 64
            buf = torch.full((h, pitch), 0xA5, dtype=torch.uint8, device="cuda")
            ms = rm.process_static(info, buf.data_ptr(), pitch)
            assert ms > 0
            got = buf.cpu().numpy()
            assert (got[:, w * 4:] == 0xA5).all(), "bytes beyond width*4 must stay untouched"
            got = got[:, :w * 4].reshape(h, w, 4)
            rm.set_info(info)
            rm.process()
            want = rm.read_pixels()
            assert np.array_equal(got, want), "call %d (%s %dx%d): differs from rm_process" % (k, name, w, h)
            ref = O.render(stock_scenes[name], w, h, info, threads=min(16, os.cpu_count() or 4))
            assert np.array_equal(got, ref["rgba"]), "call %d (%s %dx%d): differs from the oracle" % (k, name, w, h)
    finally:
        for rm in ctxs.values():
            rm.close()


def test_static_entry_point_allocates_nothing(stock_scenes):
    """The property the entry point exists for: between the device's free-memory readings (hipMemGetInfo) before
    and after a run of calls - other cameras, a second context of another size and scene taking turns on the same
    array - not a byte of device memory is taken or given back.  (The contexts and the output buffers exist
    before the first reading; one warm call per context comes first, so that anything the runtime sets up lazily
    for a first launch is not counted as the call's.)"""
    torch = pytest.importorskip("torch")
    shapes = [("chair", 640, 360), ("infinity", 333, 190)]
    ctxs, bufs = [], []
    try:
        for name, w, h in shapes:
            rm = api.Raymarching(w, h)
            rm.set_nodes(stock_scenes[name])
            ctxs.append(rm)
            bufs.append(torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda"))
        infos = [O.default_info(1 + k, k) for k in range(6)]
        for rm, buf in zip(ctxs, bufs):
            rm.process_static(infos[0], buf.data_ptr(), buf.shape[1])
        torch.cuda.synchronize()
        free0, total = torch.cuda.mem_get_info()
        for k, info in enumerate(infos):
            j = k % len(ctxs)
            ctxs[j].process_static(info, bufs[j].data_ptr(), bufs[j].shape[1])
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, "rm_process_static changed the device's free memory by %d bytes" % (free0 - free1)
        # and what the last calls wrote is the frame
        for j, (name, w, h) in enumerate(shapes):
            k = max(i for i in range(len(infos)) if i % len(ctxs) == j)
            ref = O.render(stock_scenes[name], w, h, infos[k], threads=min(16, os.cpu_count() or 4))
            assert np.array_equal(bufs[j].cpu().numpy().reshape(h, w, 4), ref["rgba"])
    finally:
        for rm in ctxs:
            rm.close()


def test_static_entry_point_limits(stock_scenes):
    """Sizes beyond the array's 131 072 tiles are refused with RM_ERR_TOO_LARGE; bad arguments are errors."""
    torch = pytest.importorskip("torch")
    info = O.default_info(1, 0)
    with api.Raymarching(7680, 4320) as rm:
        rm.set_nodes(stock_scenes["chair"])
        buf = torch.zeros(16, dtype=torch.uint8, device="cuda")
        with pytest.raises(api.RaymarchError) as e:
            rm.process_static(info, buf.data_ptr(), 7680 * 4)
        assert e.value.status == -5
    with api.Raymarching(64, 64) as rm:
        buf = torch.zeros((64, 256), dtype=torch.uint8, device="cuda")
        with pytest.raises(api.RaymarchError) as e:
            rm.process_static(info, buf.data_ptr(), 256)          # no scene yet
        assert e.value.status == -6
        rm.set_nodes(stock_scenes["chair"])
        with pytest.raises(api.RaymarchError) as e:
            rm.process_static(info, buf.data_ptr(), 255)
        assert e.value.status == -1
        rm.process_static(info, buf.data_ptr(), 256)
