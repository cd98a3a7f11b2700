#!/usr/bin/env python3
"""rebuild_summary.py TAG WORKLOAD [--out DIR]: what tools/profile_rebuild.sh TAG WORKLOAD left under gpurun_out/
as DIR/TAG.txt: the kernel trace of rank 0's side of the N>1 path, and the rebuild kernel
(rm::assemble_luminance_kernel) per launch and per frame against its algorithmic 5 B/px."""
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, HBM_PEAK_GBS

argv = sys.argv[1:]
out_dir = os.path.join(ROOT, "profiles")
if "--out" in argv:
    k = argv.index("--out")
    out_dir = os.path.join(ROOT, argv[k + 1])
    del argv[k:k + 2]
tag, workload = argv[0], argv[1]
scene, W, H = WORKLOADS[workload]
G = os.path.join(ROOT, "gpurun_out")
KERNEL = "rm::assemble_luminance_kernel"

bench_line = open(os.path.join(G, tag + "_trace_bench.json")).read().strip().splitlines()[-1]
rec = json.loads(bench_line)
fpl = int(round(rec["config"]["frames_per_launch"]))
lines = ["# %s: rank 0's side of the N>1 path on one MI355X (RM_BENCH_SOLO_GATHER=1), %s (%s.sce %dx%d), launches of %d frames" % (tag, workload, scene, W, H, fpl),
         "# three contexts render in turn; per launch one RCCL gather (of the one part, to itself) and one rm_assemble_luminance_batch",
         "# command per pass: RM_BENCH_SOLO_GATHER=1 rocprofv3 [--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 bench.py --gpus 1 --workload %s --steps 64 --warmup 32 --no-cpu-baseline" % workload,
         "# (tools/profile_rebuild.sh; every pass a run of its own)", ""]
avg_us = None
db_path = glob.glob(os.path.join(G, tag + "_trace", "**", "*.db"), recursive=True)
lines.append("## kernel trace")
lines.append("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
db = sqlite3.connect(db_path[0])
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    lines.append("%-90s %8d %14.3f %12.3f %8.3f" % (name[:90], calls, total, avg, pct))
    if name.startswith(KERNEL):
        avg_us = avg
lines += ["", "## bench line of the traced run (its rate means nothing: one rank gathers to itself)", bench_line, ""]
counters = {}
for path in sorted(glob.glob(os.path.join(G, tag + "_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(path)):
        if not r["Kernel_Name"].startswith(KERNEL):
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, by_dispatch in per.items():
        vals = [v for _, v in sorted(by_dispatch.items())]
        full = [v for v in vals if v >= 0.9 * max(vals)]           # launches of the full frame count
        counters[name] = sum(full) / len(full)
lines.append("## counters of %s, mean per launch of %d frames (sum over the chip)" % (KERNEL, fpl))
for name, v in counters.items():
    lines.append("%-28s %14.0f" % (name, v))
lines += ["", "## derived"]
px = W * H
if avg_us is not None:
    per_frame = avg_us / fpl
    alg = 5.0 * px
    lines.append("rebuild kernel: %.1f us per launch, %.2f us per %dx%d frame (average of the trace, launches of up to %d frames)" % (avg_us, per_frame, W, H, fpl))
    lines.append("algorithmic traffic 5 B/px (1 B/px of luminance in, 4 B/px of RGBA8 out) = %.1f MB per frame: %.0f GB/s = %.3f of the %.0f GB/s HBM peak"
                 % (alg / 1e6, alg / per_frame / 1e3, alg / per_frame / 1e3 / HBM_PEAK_GBS, HBM_PEAK_GBS))
if "WRITE_SIZE" in counters:
    lines.append("WRITE_SIZE %.1f MB per launch = %.2f B/px (16-byte-per-lane streaming stores: the counter reads bytes exactly, MI355X_MICROARCH.md)"
                 % (counters["WRITE_SIZE"] * 1024 / 1e6, counters["WRITE_SIZE"] * 1024 / (px * fpl)))
if "FETCH_SIZE" in counters:
    lines.append("FETCH_SIZE %.1f MB per launch as read = %.2f B/px (4-byte-per-lane loads: a width the guide does not calibrate; the planes were written by the "
                 "render kernels and gathered just before, so part of them is served by the 256 MiB Infinity Cache)" % (counters["FETCH_SIZE"] * 1024 / 1e6, counters["FETCH_SIZE"] * 1024 / (px * fpl)))
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, tag + ".txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
