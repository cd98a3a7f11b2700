"""CPU suite, world_size 2 and 8 over gloo: the N>1 path of bench.py (row bands per rank,
gather to rank 0, interleave) with the CPU restatement standing in for the render
kernel.  The assembled frame must equal the single-rank frame byte for byte, both when
RGBA8 rows travel and when the luminance planes of a launch of two frames do."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, band, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from oracle import oracle as O
    from raytracing_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nodes = O.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", "chair.sce"))
    info = O.default_info(1, 0)
    rows_max = tiling.bands_per_rank(h, band, world) * band
    buf = torch.zeros((rows_max, w, 4), dtype=torch.uint8)
    at = 0
    for y0, y1 in tiling.band_of_rank(h, band, rank, world):          # what rm_set_bands + rm_process do on a GPU
        part = O.render(nodes, w, h, info, rows=(y0, y1))["rgba"][y0:y1]
        buf[at:at + (y1 - y0)] = torch.from_numpy(part)
        at += y1 - y0
    assert at == tiling.local_rows(h, band, rank, world)
    parts = torch.zeros((world, rows_max, w, 4), dtype=torch.uint8) if rank == 0 else None
    dist.gather(buf, list(parts.unbind(0)) if rank == 0 else None, dst=0)
    if rank == 0:
        frame = torch.zeros((tiling.padded_rows(h, band, world), w, 4), dtype=torch.uint8)
        tiling.assemble_into(frame, parts, band)
        np.save(out_path, frame[:h].numpy())
    # bench.py's transport: one byte per pixel (the blue channel is the luminance), the frames of a
    # launch in one gather, RGBA8 rebuilt on rank 0 (rm_set_luminance_output / rm_assemble_luminance_batch),
    # the bands dealt by weight: rank 0, which rebuilds, holds half of what rank 1 does (rm_set_band_pattern)
    owner = tiling.deal(tiling.lighter_rank0(world, 0.5))            # 2 ranks: [1, 2]; 8 ranks: [1, 2, 2, 2, 2, 2, 2, 2]
    rows_max = tiling.max_local_rows(h, band, world, owner)
    infos = [info, O.default_info(3, 40)]
    slab = torch.zeros((len(infos), rows_max, w), dtype=torch.uint8)
    for f, inf in enumerate(infos):
        at = 0
        for y0, y1 in tiling.band_of_rank(h, band, rank, world, owner):
            part = O.render(nodes, w, h, inf, rows=(y0, y1))["rgba"][y0:y1]
            slab[f, at:at + (y1 - y0)] = torch.from_numpy(np.ascontiguousarray(part[:, :, 2]))
            at += y1 - y0
    slabs = torch.zeros((world,) + tuple(slab.shape), dtype=torch.uint8) if rank == 0 else None
    dist.gather(slab, list(slabs.unbind(0)) if rank == 0 else None, dst=0)
    if rank == 0:
        frames = [tiling.expand_luminance(tiling.assemble([slabs[r, f].numpy()[:, :, None] for r in range(world)], h, band, owner)[:, :, 0])
                  for f in range(len(infos))]
        np.save(out_path.replace(".npy", "_lum.npy"), np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,band", [(2, 96, 70, 16), (2, 64, 64, 32), (8, 64, 203, 8)])
def test_n_rank_frame_equals_one_rank_frame(tmp_path, world, w, h, band):
    """(8 ranks, 8-row bands, a last band of 3 rows: the shape of the driver's 8-GPU run, on the CPU)"""
    from oracle import oracle as O
    from raytracing_amd import tiling
    assert tiling.lighter_rank0(2, 0.5) == [1, 2] and tiling.lighter_rank0(8, 0.5) == [1] + [2] * 7
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, band, out), nprocs=world, join=True)
    nodes = O.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", "chair.sce"))
    want = O.render(nodes, w, h, O.default_info(1, 0))["rgba"]
    assert np.array_equal(np.load(out), want)
    got = np.load(out.replace(".npy", "_lum.npy"))
    assert np.array_equal(got[0], want)
    assert np.array_equal(got[1], O.render(nodes, w, h, O.default_info(3, 40))["rgba"])
