#!/usr/bin/env python3
"""Headline benchmark: Mrays/s rendering Scenes/chair.sce at 3840x2160 (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame of the hot path: rm_set_info + one launch of the fused
raygen/march/shadow/shade kernel over every pixel (no work is skipped or cached;
every step re-renders the frame).  Scene table and camera block are resident on
the device before the timed region; the framebuffer stays in HBM.

N > 1: the frame is cut into row bands dealt round-robin to the ranks
(raytracing_amd/tiling.py); each rank renders its bands, then one RCCL gather per
frame brings them to rank 0, which interleaves them back into the frame.  The
gather of frame k runs on a second stream under the render of frame k+1.  The
total work is the same frame whatever N is: scaling = "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (scene, width, height)
    "chair_4k": ("chair", 3840, 2160),
    "chair_1080p": ("chair", 1920, 1080),
    "infinity_4k": ("infinity", 3840, 2160),
    "csg1024_4k": ("csg1024", 3840, 2160),
    "chair_8k": ("chair", 7680, 4320),
}


def load_nodes(scene):
    from raytracing_amd import api
    return api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))


def golden_frame(scene, w, h):
    with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
        return json.load(f).get("%s_%dx%d" % (scene, w, h))


def cpu_baseline(scene, w, h, info, want):
    """The CPU restatement (oracle/) timed on this box's host cores, one full frame."""
    from oracle import oracle as O
    nodes = O.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    if scene == "csg1024":          # a full 4K frame of this scene is ~an hour of CPU: sample rows
        rows = (h // 2 - 2, h // 2 + 2)
        t0 = time.perf_counter()
        O.render(nodes, w, h, info, rows=rows)
        secs = time.perf_counter() - t0
        return {"value": w * (rows[1] - rows[0]) / secs / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "rows %d-%d of the %dx%d frame, 1 thread" % (rows[0], rows[1] - 1, w, h)}
    r = O.render(nodes, w, h, info, threads=threads, want_counters=True)
    out = {"value": w * h / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": "1 full %dx%d frame of %s.sce, rows dealt over %d threads" % (w, h, scene, threads),
           "seconds": r["seconds"], "flops": r["counters"]["flops"], "hits": r["counters"]["hits"]}
    if want is not None:
        out["matches_golden_counters"] = (r["counters"]["flops"] == want["counters"]["flops"] and
                                          r["counters"]["hits"] == want["hits"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="chair_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band for N>1 (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--norm-batch", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import api, tiling
    from oracle import oracle as O      # camera block + cpu_baseline only; never in the timed path

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the render path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene, w, h = WORKLOADS[args.workload]
    nodes = load_nodes(scene)
    info = O.default_info(1, 0)          # main.cpp default camera, light at +1 degree ("frame 0")
    want = golden_frame(scene, w, h)

    rm = api.Raymarching(w, h, device=local_rank)
    rm.set_nodes(nodes)
    rm.set_info(info)
    if args.norm_batch:
        rm.set_norm_batch(args.norm_batch)
    render_stream = torch.cuda.Stream(device=dev)
    rm.set_stream(render_stream.cuda_stream)

    band = 0
    if world > 1:
        band = args.band_rows or tiling.default_band_rows(h, world)
        rm.set_bands(band, rank, world)
        rows_max = tiling.bands_per_rank(h, band, world) * band
        bufs = [torch.zeros((rows_max, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        comm_stream = torch.cuda.Stream(device=dev)
        rendered = [torch.cuda.Event() for _ in range(2)]
        gathered = [torch.cuda.Event() for _ in range(2)]
        parts = [torch.zeros((world, rows_max, w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
                 for _ in range(2)]
        frame_padded = (torch.zeros((tiling.padded_rows(h, band, world), w, 4), dtype=torch.uint8, device=dev)
                        if rank == 0 else None)
        frame = frame_padded[:h] if rank == 0 else None
    else:
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        rm.set_output(frame.data_ptr(), w * 4)

    def step(k):
        rm.set_info(info)
        if world == 1:
            rm.process_async()
            return
        b = k & 1
        render_stream.wait_event(gathered[b])          # buffer b is free again
        rm.set_output(bufs[b].data_ptr(), w * 4)
        rm.process_async()
        rendered[b].record(render_stream)
        comm_stream.wait_event(rendered[b])
        with torch.cuda.stream(comm_stream):
            dist.gather(bufs[b], list(parts[b].unbind(0)) if rank == 0 else None, dst=0)
            if rank == 0:
                tiling.assemble_into(frame_padded, parts[b], band)
            gathered[b].record(comm_stream)

    def drain():
        render_stream.synchronize()
        if world > 1:
            comm_stream.synchronize()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(render_stream)
    for k in range(args.steps):
        step(args.warmup + k)
    ev1.record(render_stream)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)     # average launch-to-launch time on the render stream

    if world > 1:
        tmax = torch.tensor([secs], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        secs = float(tmax.item())

    if rank == 0:
        got = frame.cpu().numpy()
        hits = int(got[:, :, :3].any(-1).sum())
        import hashlib
        sha = hashlib.sha256(got.tobytes()).hexdigest()
        mrays = w * h * args.steps / secs / 1e6
        out = {
            "metric": "Mrays/s at 3840x2160 on chair.sce" if args.workload == "chair_4k" else "Mrays/s (%s)" % args.workload,
            "value": mrays, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": secs * 1e3 / args.steps,
            "frames_per_s": args.steps / secs,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s.sce %dx%d, main.cpp default camera, light +1 deg, one primary ray per pixel" % (scene, w, h),
                       "scene_nodes": int(len(nodes)), "partition": "whole frame" if world == 1 else
                       "row bands of %d dealt round-robin over %d ranks, RCCL gather to rank 0" % (band, world)},
            "parity": {"hit_pixels": hits, "rgba_sha256": sha,
                       "golden_hit_pixels": want["hits"] if want else None,
                       "frame_matches_golden": (sha == want["rgba_sha256"]) if want else None},
            "kernel_ms": kernel_ms,
        }
        if want:
            flops = want["counters"]["flops"] / world      # per launch: each rank renders 1/N of the frame
            achieved = flops / (kernel_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "valu_fp32", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / FP32_VALU_PEAK_TFLOPS, "traffic": None,
                               "algorithmic_flops_per_ray": want["flops_per_ray"],
                               "note": "algorithmic FLOPs counted by the CPU restatement (SURVEY.md appendix D); "
                                       "fp32 adds and multiplies are not contracted, so the attainable ceiling is "
                                       "half the FMA peak",
                               "hbm": {"achieved": w * h * 4 / world / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "bytes_per_ray": 4}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, w, h, info, want)
            out["stats"] = rm.scene_stats()
        print(json.dumps(out), flush=True)

    rm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
