#!/usr/bin/env python3
"""Headline benchmark: Mrays/s rendering Scenes/chair.sce at 3840x2160 (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame of the hot path: rm_set_info + one launch of the fused
raygen/march/shadow/shade kernel over every pixel (no work is skipped or cached;
every step re-renders the frame; the only thing a frame takes from the one
before it is the ORDER in which its tiles are handed out, longest first).  Scene
table and camera block are resident on the device before the timed region; the
framebuffer stays in HBM.

N = 1: one context, one stream, launches of up to 32 frames back to back (so the kernel's duration
under rocprofv3 is the launch's time).  A timed run is --repeats blocks of exactly --steps frames each,
queued straight behind an untimed block of the same shape and behind each other; the MEDIAN block is
`ms_per_step` and `value`, the others and every launch's own duration are in the line.
--in-flight F renders with F contexts taking launches in turn, each on its own stream (no gain since
launches hold 20-32 frames).

N > 1: the frame is cut into row bands dealt round-robin to the ranks
(raytracing_amd/tiling.py); each rank renders its bands, then one RCCL gather per
frame brings them to rank 0, which interleaves them back into the frame.  The
gather of frame k runs on a second stream under the render of the next frames,
and three frames are in flight per rank (a rank's share of a frame is small and
its time is set by the longest tile, not by the pixel count).  The total work is
the same frame whatever N is: scaling = "strong".

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
# what the vector unit issues when nothing may be contracted into an FMA (every add and multiply
# its own instruction): tools/microbench/valu_peak.hip, v_add_f32 at 8 waves per SIMD
FP32_VALU_NO_FMA_TFLOPS = 63.0
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (scene, width, height)
    "chair_4k": ("chair", 3840, 2160),
    "chair_1080p": ("chair", 1920, 1080),
    "infinity_4k": ("infinity", 3840, 2160),
    "csg1024_4k": ("csg1024", 3840, 2160),
    "chair_8k": ("chair", 7680, 4320),
    # BASELINE.json configs[4]: the 120-frame camera sweep (tests/golden/sweep_rays_info.bin: frame k =
    # k+1 light rotations of 1 degree and a right-button drag of k pixels, Source/main.cpp:248-266)
    "chair_8k_sweep": ("chair", 7680, 4320),
}


def load_nodes(scene):
    from raytracing_amd import api
    return api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))


def golden_frame(scene, w, h):
    with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
        return json.load(f).get("%s_%dx%d" % (scene, w, h))


def measured_traffic(workload, frames_per_launch):
    """HBM bytes per launch from the latest committed counter profile of THIS workload (rocprofv3 --pmc
    cannot run inside the timed region): {"bytes": ..., "source": ...} or None.  A profile taken at
    another number of frames per launch is scaled to this run's."""
    import glob
    import re
    names = glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json"))
    for path in sorted(names, key=lambda p: [int(v) for v in re.findall(r"\d+", os.path.basename(p))], reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get("workload") != workload:
                continue
            per = rec.get("frames_per_launch", 8)
            return {"bytes": rec["hbm_bytes_per_launch"] * frames_per_launch / per, "source": rec["source"]}
        except (OSError, ValueError, KeyError):
            continue
    return None


def host_cpu():
    """What the CPU baseline ran on: the model name, the machine's logical CPUs, and how many this
    process may use (the GPU box gives a one-GPU job a share of the machine)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    return model, os.cpu_count() or 1, allowed


def cpu_baseline(scene, w, h, info, want):
    """The CPU restatement (oracle/) timed on this box's host cores, one full frame."""
    from oracle import oracle as O
    nodes = O.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", scene + ".sce"))
    # every core this job may use: the GPU box hands a one-GPU job 16 threads of the machine and says so in
    # OMP_NUM_THREADS (its nproc shows the whole machine); elsewhere, the cores the process may run on
    model, nproc, allowed = host_cpu()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or allowed
    host = {"cpu_model": model, "nproc": nproc, "cpus_allowed": allowed,
            "threads_from": "OMP_NUM_THREADS" if os.environ.get("OMP_NUM_THREADS") else "sched_getaffinity"}
    if scene == "csg1024":          # a full 4K frame of this scene is ~an hour of CPU: sample rows
        rows = (h // 2 - 2, h // 2 + 2)
        t0 = time.perf_counter()
        O.render(nodes, w, h, info, rows=rows)
        secs = time.perf_counter() - t0
        return {"value": w * (rows[1] - rows[0]) / secs / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "rows %d-%d of the %dx%d frame, 1 thread" % (rows[0], rows[1] - 1, w, h), **host}
    r = O.render(nodes, w, h, info, threads=threads, want_counters=True)
    out = {"value": w * h / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": "1 full %dx%d frame of %s.sce, rows dealt over %d threads" % (w, h, scene, threads),
           "seconds": r["seconds"], "flops": r["counters"]["flops"], "hits": r["counters"]["hits"], **host}
    # one thread, on a band of 32 rows through the middle of the image (SURVEY.md 8d asks for both)
    rows = (h // 2 - 16, h // 2 + 16)
    t0 = time.perf_counter()
    O.render(nodes, w, h, info, rows=rows)
    out["one_thread"] = {"value": w * (rows[1] - rows[0]) / (time.perf_counter() - t0) / 1e6, "unit": "Mrays/s",
                         "sample": "rows %d-%d of the same frame" % (rows[0], rows[1] - 1)}
    if want is not None:
        out["matches_golden_counters"] = (r["counters"]["flops"] == want["counters"]["flops"] and
                                          r["counters"]["hits"] == want["hits"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=7,
                    help="timed blocks of --steps frames, back to back; the median block is reported (>= 1)")
    ap.add_argument("--workload", default="chair_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band for N>1 (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the extra frame-per-launch measurement (profiling runs)")
    ap.add_argument("--in-flight", type=int, default=0, help="contexts taking launches in turn (0 = 1 for N=1, 3 for N>1)")
    ap.add_argument("--rank0-share", type=float, default=None,
                    help="N>1: rank 0's bands as a fraction of another rank's (it also rebuilds the frames); default: measured")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="most frames whose tiles one launch pools (rm_set_batch), 1..32; 0 = 16 on one GPU, 32 on several")
    ap.add_argument("--warmup-seconds", type=float, default=0.25,
                    help="warm up for at least this long as well as for --warmup frames (clocks, tile order)")
    args = ap.parse_args()

    # N>1: four contexts render a rank's shares of consecutive frames at once, each on its own stream,
    # beside the gather's two streams.  HIP maps streams onto 4 hardware queues by default and streams
    # that share one run in turn (measured, tools/slice_rate.py: 4 in flight 9.9 k band-sets/s on 4
    # queues, 16.7 k on 8), so ask for 8 before the runtime starts.
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from raytracing_amd import api, tiling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the render path")
    # Rehearsal on a box with one GPU (RM_BENCH_REHEARSAL=1): every rank renders on device 0 and the
    # gather goes through gloo on host copies.  Same step logic, same band layout, same assembly; the
    # numbers it prints mean nothing.
    rehearsal = os.environ.get("RM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    # RM_BENCH_SOLO_GATHER=1 with --gpus 1: the N>1 code path with one rank - bands, luminance planes, launches in
    # turn on three contexts and a real RCCL gather (of one part, to itself) - so that a one-GPU box can at least
    # check that the collective takes the tensors this path hands it.  The numbers it prints mean nothing.
    solo = world == 1 and os.environ.get("RM_BENCH_SOLO_GATHER", "0") == "1"
    multi = world > 1 or solo
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if not rehearsal and torch.cuda.device_count() < local_world:
        raise SystemExit("bench.py: %d ranks on this node but only %d GPU(s) visible: one process per GPU (set "
                         "RM_BENCH_REHEARSAL=1 to rehearse several ranks on one GPU over gloo)" % (local_world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if solo:
            os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene, w, h = WORKLOADS[args.workload]
    nodes = load_nodes(scene)
    # main.cpp's default camera with the light at +1 degree ("frame 0") is block 0 of the committed sweep
    # (tests/test_oracle.py checks that it is); the other blocks are config 5's camera path
    blocks = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    info = blocks[0:1].copy()
    sweep = blocks if args.workload.endswith("_sweep") else None
    want = None if sweep is not None else golden_frame(scene, w, h)
    in_flight = args.in_flight or (1 if not multi else 3)

    # one context per frame in flight, each on a stream of its own
    ctxs, streams = [], []
    for _ in range(in_flight):
        rm = api.Raymarching(w, h, device=local_rank)
        rm.set_nodes(nodes)
        rm.set_info(info)
        st = torch.cuda.Stream(device=dev)
        rm.set_stream(st.cuda_stream)
        ctxs.append(rm)
        streams.append(st)

    band = 0
    # Frames per launch (rm_set_batch): up to 32 (RM_MAX_BATCH), and a timed run of K frames goes out in the fewest
    # launches of nearly equal size (FrameBatcher.plan).  The reference renders a frame per Process() call;
    # here the tiles of consecutive frames of the sequence form one pool for the persistent wavefronts, so the few
    # long tiles of a frame (one ray marching ~500 steps is 0.2 ms however idle the chip is) run beside the short
    # tiles of the others.  At N>1 it is what makes a rank's launch worth launching: an 8-rank share of ONE frame
    # is too little work to fill the chip for as long as its longest ray marches.  `single_frame_launch` in the
    # output is the frame-at-a-time figure (the reference's own call pattern).
    # N=1, chair 4K, same box: 0.1519 ms per frame in launches of 16, 0.1506 of 20, 0.1500 of 29; the driver's 20 steps
    # are one launch (0.152; as two of 10: 0.154-0.155).
    # N>1: a rank's launch at N=8 then holds four frames' worth of tiles (tools/slice_rate.py, render
    # side only, 3 contexts in flight: 48 k shares/s per GPU at 4K against 6.4 k whole frames/s on one GPU, 11.1 k
    # against 1.43 k at 8K), and at least two launches, so that a gather runs under the next launch's kernel.
    # A launch of more than ~8 M tiles gains nothing (8K sweep, same box: 0.346-0.348 ms per frame in launches of 8 or
    # 16, 0.354 of 32 - and on a moving camera the tile order a frame inherits is the older the longer the launch).
    tiles_per_rank = ((w + 7) // 8) * ((h + 7) // 8) / max(1, world)
    batch = args.frames_per_launch or max(1, min(32, int((1 << 23) / tiles_per_rank)))
    batch = max(1, min(32, batch))
    # the timed run's launch size is the warm-up's too (70 steps at up to 32 per launch: 24, 23 and 23, and the
    # warm-up's launches hold 24): a launch learns its tile order and its grid from the launch before it
    if args.steps > 0:
        n_launches = max(-(-args.steps // batch), min(2, args.steps) if multi else 1)
        # launches of ONE size where a few more of them allow it (200 steps at up to 32: 8 of 25, not 4 of 29 and 3 of
        # 28): the timed blocks repeat, and a launch of more frames than the one before it finds the tiles of the
        # extra frames in no list (they go out last, in natural order: 29 behind 28 measured 6.7 ms against 4.5)
        for n in range(n_launches, n_launches + n_launches // 2 + 1):
            if args.steps % n == 0:
                n_launches = n
                break
        batch = -(-args.steps // n_launches)
    weights, deal_note = None, None
    if multi:
        # Every rank renders its bands of a frame; what travels to rank 0 is the frame's luminance plane
        # (1 B/px: a pixel is a function of that byte, include/raymarching_amd.h), the frames of a launch
        # in one RCCL gather, and rank 0 rebuilds the RGBA8 frames of the launch from the planes of all ranks in
        # one kernel (rm_assemble_luminance_batch).  At 4 B/px and 8 ranks the gather alone (29 MB per frame into
        # one GPU's xGMI links) would bound the frame rate below what the kernels deliver.
        band = args.band_rows or tiling.default_band_rows(h, world)
        # Rank 0 also writes every frame's 4 B/px (5 B/px of HBM traffic a frame: 9 us at 4K and 38 us at 8K alone on
        # the chip, more beside its own render kernels - an even share of the rendering is 20 and 45 us), so it is
        # dealt fewer bands (rm_set_band_pattern).  How many fewer is measured below, before the warm-up
        # (--rank0-share overrides it: 1 = an even deal).
        comm_stream = torch.cuda.Stream(device=dev)
        rendered = [torch.cuda.Event() for _ in ctxs]
        gathered = [torch.cuda.Event() for _ in ctxs]
        # the frames of the launch gathered last, rebuilt (gathers and rebuilds run in order on one stream)
        frames_out = torch.zeros((batch, h, w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        frame = None
        frames = None
        lay = {}

        def lay_out(owner):
            """Bands dealt as `owner` says; slabs and parts sized for that deal (what the gather moves)."""
            rows = -(-tiling.max_local_rows(h, band, world, owner) // 16) * 16
            lay["rows_max"], lay["bufs"], lay["parts"] = rows, [], []
            for rm in ctxs:
                rm.set_bands(band, rank, world, owner)
                lay["bufs"].append(torch.zeros((batch, rows, w), dtype=torch.uint8, device=dev))
                rm.set_luminance_output(lay["bufs"][-1].data_ptr(), w)
                lay["parts"].append(torch.zeros((world, batch, rows, w), dtype=torch.uint8, device=dev) if rank == 0 else None)
                rm.set_batch(batch, 0, rows * w)       # RGBA8 into the context's own planes, luminance into the slab

        def rebuild(j, n):
            rows = lay["rows_max"]
            ctxs[j].assemble_luminance_batch(lay["parts"][j].data_ptr(), batch * rows * w, rows * w, w, n,
                                             frames_out.data_ptr(), w * 4, h * w * 4, stream=comm_stream.cuda_stream)

        weights = tiling.lighter_rank0(world, 1.0 if args.rank0_share is None else args.rank0_share)
        lay_out(tiling.deal(weights))
        lay["gather"] = True
    else:
        frames = [torch.zeros((batch, h, w, 4), dtype=torch.uint8, device=dev) for _ in ctxs]
        for rm, f in zip(ctxs, frames):
            rm.set_output(f.data_ptr(), w * 4)
        frame = None                                   # the last frame rendered: set by launch()
        for rm in ctxs:                                # room for `batch` frames per launch, once, before anything is timed
            rm.set_batch(batch, h * w * 4, 0)
    def before_launch(j):
        if multi:
            streams[j].wait_event(gathered[j])         # context j's slab is free again

    host_times = [] if os.environ.get("RM_BENCH_HOST_TIMES") else None     # seconds of host time per after_launch
    launch_marks = None                                # set for the timed part
    comm_marks = None                                  # N>1, the timed part: events around every launch's gather and rebuild
    incomplete = []                                    # RM_ERR_INCOMPLETE met at a drain (reported, and the run exits non-zero)

    def after_launch(j, n):
        """N>1: the launch's luminance planes to rank 0 in one gather, RGBA8 frames rebuilt there."""
        if not multi:
            if launch_marks is not None:               # the timed part, one stream: an event behind every launch
                e = torch.cuda.Event(enable_timing=True)
                e.record(streams[j])
                launch_marks.append(e)
            return
        t_host = time.perf_counter()
        rendered[j].record(streams[j])
        comm_stream.wait_event(rendered[j])
        with torch.cuda.stream(comm_stream):           # same order of gathers on every rank
            if comm_marks is not None:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(comm_stream)
            if not lay["gather"]:
                pass                                   # the deal is being measured: render and rebuild only
            elif rehearsal:
                comm_stream.synchronize()
                host = lay["bufs"][j][:n].cpu()
                got = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, got, dst=0)
                if rank == 0:
                    lay["parts"][j][:, :n].copy_(torch.stack(got))
            else:
                dist.gather(lay["bufs"][j][:n], list(lay["parts"][j][:, :n].unbind(0)) if rank == 0 else None, dst=0)     # the n frames of this launch only
            if comm_marks is not None:
                ev[1].record(comm_stream)
            if rank == 0:
                rebuild(j, n)
            if comm_marks is not None:
                ev[2].record(comm_stream)
                comm_marks.append(ev)
            gathered[j].record(comm_stream)
        if host_times is not None:
            host_times.append(time.perf_counter() - t_host)

    from raytracing_amd.batching import FrameBatcher
    run = FrameBatcher(ctxs, batch, before=before_launch, after=after_launch)

    def step(k):
        run.step(info if sweep is None else sweep[k % len(sweep):k % len(sweep) + 1])

    def drain():
        run.flush()                                    # frames that do not fill a launch
        for rm in ctxs:
            try:
                rm.synchronize()                       # also the completeness check: a frame with tiles missing...
            except api.RaymarchError as e:
                if e.status != -10:                    # ...(RM_ERR_INCOMPLETE) is recorded, the ranks stay in step, and the
                    raise                              # run ends with the ranks named and a non-zero exit code
                incomplete.append("%s: tiles finished %d, launched %d, watchdog flag %d" % ((str(e),) + rm.last_check()))
        if multi:
            comm_stream.synchronize()
        torch.cuda.synchronize()

    if multi and world > 1 and args.rank0_share is None:
        # The deal.  For each candidate share of rank 0 every rank runs its part of the job without the collective
        # - its bands rendered in launches on its contexts in turn, rank 0 rebuilding every launch's frames beside
        # them - and reports frames per second; the job runs at its slowest rank's rate, and the candidate where
        # that is highest is taken.  (Measured rather than modelled: rank 0's time is not its share times a rate plus
        # the rebuild's time alone - a launch is never shorter than its longest ray, and the two kernels share the
        # chip.  One GPU, tools/slice_rate.py, 8 ranks at 4K: even deal 33 k frames/s on rank 0 against 45 k on the
        # others, half a share 38 k against 42 k, a quarter share 55 k against 44 k.)
        shares = (1.0, 0.875, 0.75, 0.5, 0.375, 0.25, 0.125)
        lay["gather"] = False
        table = []
        for share in shares:
            lay_out(tiling.deal(tiling.lighter_rank0(world, share)))
            for k in range(2 * batch * in_flight):     # the tile order and the grid are learnt from the launch before
                step(k)
            drain()
            t_deal, done = time.perf_counter(), 0
            while done < 3 * batch * in_flight or time.perf_counter() - t_deal < 0.04:     # no collective: ranks need not agree
                for k in range(batch * in_flight):
                    step(k)
                done += batch * in_flight
            drain()
            table.append(done / (time.perf_counter() - t_deal))
        lay["gather"] = True
        rates = torch.tensor(table, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(rates, op=dist.ReduceOp.MIN)
        rates = [float(x) for x in rates.cpu()]
        best = max(range(len(shares)), key=lambda k: rates[k])
        weights = tiling.lighter_rank0(world, shares[best])
        lay_out(tiling.deal(weights))
        deal_note = {"rank0_share_tried": list(shares), "slowest_rank_frames_per_s_without_gather": rates,
                     "this_rank_frames_per_s_without_gather": table}

    # Warm-up: at least --warmup frames, in whole launches, and at least --warmup-seconds of them: a fresh
    # box needs a few hundred milliseconds of work before its clocks are where a long run has them, and the
    # tile order is learnt from the launch before.  Every rank runs the same number of launches (the
    # gathers are collective), so the ranks agree on the count.
    import gc
    gc.collect()                                       # now, while nothing is timed: the chip must not sit idle between
    gc.disable()                                       # the warm-up and the timed blocks, and no collector pause falls into them
    warm = -(-args.warmup // batch) * batch
    t_warm = time.perf_counter()
    for k in range(warm):
        step(k)
    drain()
    while True:
        more = 1 if time.perf_counter() - t_warm < args.warmup_seconds else 0
        if multi:
            flag = torch.tensor([more], dtype=torch.int32, device="cpu" if rehearsal else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = int(flag.item())
        if not more:
            break
        for k in range(4 * batch * in_flight):
            step(warm + k)
        warm += 4 * batch * in_flight
        drain()
    # The timed part (raytracing_amd/batching.py run_blocks): the collector ran before the warm-up; behind
    # the barrier and synchronize the contract asks for, one block of --steps frames that is NOT timed and
    # --repeats timed blocks of exactly --steps frames each, every block queued straight behind the one before -
    # no wait, no collector, no allocation between them - and the barrier and synchronize again.  A block's time
    # is taken on the device, between the events that follow its last launch and the previous block's (on every
    # stream that carries its work; the latest one counts), the MAX over ranks per block, and `ms_per_step` /
    # `value` are the MEDIAN block's.  Round 2 timed one block behind a drain: the chip had been idle for the
    # milliseconds of the drain, the collector and the barrier, and that one launch read 17 % slower on the
    # driver's box than the same launch in a row of them (`first_block_ms` is the block that
    # starts on the idle chip, reported so that the effect stays visible).
    from raytracing_amd.batching import run_blocks
    if multi:
        dist.barrier()
    torch.cuda.synchronize()

    timed_streams = list(streams) + ([comm_stream] if multi else [])
    launch_marks = [] if in_flight == 1 and not multi else None      # one stream: an event behind every launch
    comm_marks = [] if multi else None

    def mark():
        evs = []
        for st in timed_streams:
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            evs.append(e)
        return evs

    def info_of(k):
        return info if sweep is None else sweep[k % len(sweep):k % len(sweep) + 1]

    run.launches = 0
    t0 = time.perf_counter()
    start = mark()
    marks, _ = run_blocks(run, info_of, args.steps, args.repeats, mark, first=warm)
    drain()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gc.enable()
    launches = max(1, run.launches // (args.repeats + 1))            # per block
    if host_times:
        print("host time per after_launch (gather + rebuild), last %d: median %.1f us" % (
            launches, 1e6 * sorted(host_times[-launches:])[launches // 2]), file=sys.stderr)
    # when each block ended, in ms after `start` (the latest of its streams)
    ends = [max(s0.elapsed_time(e) for s0, e in zip(start, evs)) for evs in marks]
    block_ms = [b - a for a, b in zip(ends[:-1], ends[1:])]         # the timed blocks; ends[0] closes the untimed one
    first_block_ms = ends[0]
    if multi:
        tmax = torch.tensor(block_ms + [first_block_ms], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        block_ms, first_block_ms = [float(x) for x in tmax[:-1].cpu()], float(tmax[-1])
    order = sorted(range(len(block_ms)), key=lambda r: block_ms[r])
    mid = order[(len(order) - 1) // 2]                               # the median block (the lower one of two)
    secs = block_ms[mid] * 1e-3
    # One context: every launch's own duration (event behind it against the event behind the one before).  With
    # several in flight kernels overlap; the block's time per launch is what counts then.
    kernel_ms_all = None
    if launch_marks is not None:
        edges = [start[0]] + launch_marks
        kernel_ms_all = [a.elapsed_time(b) for a, b in zip(edges[:-1], edges[1:])]
        mine = kernel_ms_all[(mid + 1) * launches:(mid + 2) * launches]
        kernel_ms = sum(mine) / len(mine)
    else:
        kernel_ms = block_ms[mid] / launches
    if not multi:
        frame = frames[run.last[0]][run.last[1]]
    elif rank == 0:
        frame = frames_out[run.last[1]]

    # N>1: what every rank saw, so that the first run on a real 8-GPU node explains itself - the device each rank
    # bound, its render-only rate for every candidate deal, the time its launches' gathers and rebuilds took on the
    # communication stream (rank 0's gather is the ingest of all parts; another rank's is its send), and whether
    # any of its launches came back incomplete
    per_rank = None
    if multi:
        def med(v):
            return sorted(v)[len(v) // 2] if v else None
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(),
                "device_name": props.name, "device_uuid": str(getattr(props, "uuid", "")),
                "gather_ms_per_launch_median": med([e[0].elapsed_time(e[1]) for e in comm_marks]),
                "gather_ms_per_launch_max": max([e[0].elapsed_time(e[1]) for e in comm_marks], default=None),
                "rebuild_ms_per_launch_median": med([e[1].elapsed_time(e[2]) for e in comm_marks]) if rank == 0 else None,
                "render_only_frames_per_s_by_rank0_share": (deal_note or {}).pop("this_rank_frames_per_s_without_gather", None),
                "incomplete": incomplete}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        got = frame.cpu().numpy()
        hits = int(got[:, :, :3].any(-1).sum())
        import hashlib
        sha = hashlib.sha256(got.tobytes()).hexdigest()
        mrays = w * h * args.steps / secs / 1e6
        out = {
            "metric": "Mrays/s at 3840x2160 on chair.sce" if args.workload == "chair_4k" else "Mrays/s (%s)" % args.workload,
            "value": mrays, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_frames_run": warm,
            "ms_per_step": secs * 1e3 / args.steps,
            "frames_per_s": args.steps / secs,
            # --repeats timed blocks of --steps frames each, back to back behind an untimed one: the median block is
            # `ms_per_step` and `value`; the others, the block that started on the idle chip, and the wall clock over
            # all repeats + 1 blocks (barrier and synchronize on both sides) are here to check it against
            "repeats": args.repeats, "block_ms": block_ms,
            "ms_per_step_min": min(block_ms) / args.steps, "ms_per_step_max": max(block_ms) / args.steps,
            "first_block_ms": first_block_ms,
            "wall_ms_per_step_all_blocks": wall * 1e3 / (args.steps * (args.repeats + 1)),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, **({"rehearsal": True} if rehearsal else {}),
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s.sce %dx%d, main.cpp default camera, light +1 deg, one primary ray per pixel" % (scene, w, h),
                       "scene_nodes": int(len(nodes)), "contexts_in_flight": in_flight, "frames_per_launch": args.steps / launches, "frames_per_launch_max": batch,
                       "partition": "whole frame" if not multi else
                       "row bands of %d dealt to %d ranks %s; the %d frames of a launch gathered together to rank 0 "
                       "as luminance planes (1 B/px, RCCL), RGBA8 frames rebuilt there in one kernel" % (
                           band, world, "evenly" if len(set(weights)) == 1 else "by weight (rank 0 also rebuilds the frames)", batch),
                       **({"band_weights": weights} if multi else {}), **({"band_deal_measured": deal_note} if deal_note else {})},
            "parity": {"hit_pixels": hits, "rgba_sha256": sha,
                       "golden_hit_pixels": want["hits"] if want else None,
                       "frame_matches_golden": (sha == want["rgba_sha256"]) if want else None},
            "kernel_ms": kernel_ms, "launches": launches,
            **({"kernel_ms_per_launch": kernel_ms_all} if kernel_ms_all else {}),
        }
        if multi:
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = dist.get_backend()
            out["ranks"] = per_rank
            out["incomplete_ranks"] = [r["rank"] for r in per_rank if r["incomplete"]]
            out["distinct_devices"] = len({(r["device"], r["device_uuid"]) for r in per_rank})
        if want:
            # FLOPs of one launch (1/N of each of its frames), counted by the CPU restatement.
            # The kernel marches a primary ray only where it can be near the scene's volume (ray windows,
            # include/raymarching_amd.h): the FLOPs it EXECUTES are those of the windowed march, restated and
            # counted in the oracle too (tests/golden/frames.json "windowed"), and they, not the reference march's,
            # are what `achieved` and `frac` are made of.  `reference_march` prices the same frames at the
            # reference's work (every ray marched to RM_MAX_DIST): the rate a kernel without windows would have to
            # sustain for this frame time - a speed-up figure, not a utilisation (it can exceed the peak).
            # Since round 3 the kernel also shades a hit whose surface faces away from the light without marching its
            # shadow ray (the reference multiplies that march's result by zero, Source/cuda_kernels.cu:694): the
            # "windowed" counters restate that too, and the shadow windows: chair.sce 4K 631 -> 532 FLOP per ray.  (With RM_RAY_WINDOWS=0,
            # a diagnostic, the reference march's count is used and overstates what the kernel does by that much.)
            windows = os.environ.get("RM_RAY_WINDOWS", "1") != "0" and "windowed" in want
            done = want["windowed"] if windows else want
            per_launch = 1.0 / world * args.steps / launches
            achieved = done["counters"]["flops"] * per_launch / (kernel_ms * 1e-3) / 1e12
            reference = want["counters"]["flops"] * per_launch / (kernel_ms * 1e-3) / 1e12
            traffic = measured_traffic(args.workload, args.steps / launches) if not multi else None
            out["roofline"] = {"bound": "valu_fp32", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                               "traffic": traffic["bytes"] if traffic else None,
                               "traffic_source": traffic["source"] if traffic else None,
                               "algorithmic_flops_per_ray": done["flops_per_ray"], "ray_windows": windows,
                               "peak_no_fma": FP32_VALU_NO_FMA_TFLOPS, "frac_of_no_fma_peak": achieved / FP32_VALU_NO_FMA_TFLOPS,
                               "reference_march": {"flops_per_ray": want["flops_per_ray"], "achieved": reference,
                                                   "frac": reference / FP32_VALU_PEAK_TFLOPS},
                               "note": "the path is bound by fp32 vector issue, not by HBM or MFMA (SURVEY.md 8d). "
                                       "achieved = FLOPs the kernel executes in one launch (the product's march - ray windows, no shadow march for unlit hits - counted by "
                                       "the CPU restatement; SURVEY.md appendix D: add/sub/mul/div/sqrt/floor/min/max = 1) "
                                       "over the kernel's average duration; peak = 157.3 TFLOP/s of FMAs.  reference_march: "
                                       "the same frames priced at the FLOPs of the reference's march, which takes every ray "
                                       "to RM_MAX_DIST - not a utilisation.  Results must match "
                                       "the reference bit for bit, so nothing is contracted into an FMA and sqrt and "
                                       "division are correctly rounded: peak_no_fma is what the chip issues then "
                                       "(measured: tools/microbench/valu_peak.hip)"
                                       + ("; csg1024: the kernel also skips union members behind bounding-sphere tests, which "
                                          "the count does not see - an upper bound of the work done" if scene == "csg1024" else "")
                                       + ("; chair: the generated code also leaves out the carved spheres' square roots in trips where a "
                                          "wave-wide test proves that they cannot decide the max (jit.cpp bound_skip, sphere_skip), which "
                                          "the count does not see - an upper bound of the work done" if scene == "chair" else ""),
                               "hbm": {"achieved": w * h * 4 / world * args.steps / launches / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "bytes_per_ray": 4}}
        if not multi and batch > 1 and not args.no_single_frame:
            # the same frame one per launch, as the reference's Process() renders it
            with api.Raymarching(w, h, device=local_rank) as one:
                one.set_nodes(nodes)
                one.set_info(info if sweep is None else sweep[0:1])
                for _ in range(10):
                    one.process_async()
                one.synchronize()
                ms = sorted(one.process() for _ in range(30))[15]
            out["single_frame_launch"] = {"kernel_ms": ms, "Mrays/s": w * h / ms / 1e3}
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, w, h, info, want)
            out["stats"] = ctxs[0].scene_stats()
        print(json.dumps(out), flush=True)

    for rm in ctxs:
        rm.close()
    failed = bool(incomplete)
    if multi:
        failed = any(r["incomplete"] for r in per_rank)
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: a launch came back RM_ERR_INCOMPLETE (tiles missing or a wavefront stopped by its watchdog)")


if __name__ == "__main__":
    main()
