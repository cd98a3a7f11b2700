#!/usr/bin/env python3
"""profile_summary.py TAG [WORKLOAD] [note] [--out DIR]: turn what tools/profile_round.sh TAG WORKLOAD left under
gpurun_out/ into DIR/TAG.txt (kernel trace, bench line, counters per launch, derived figures),
DIR/TAG_counters.json (read by bench.py for roofline.traffic of that workload) and DIR/TAG_bench.json.
DIR defaults to profiles/."""
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS

argv = sys.argv[1:]
out_dir = os.path.join(ROOT, "profiles")
if "--out" in argv:
    k = argv.index("--out")
    out_dir = os.path.join(ROOT, argv[k + 1])
    del argv[k:k + 2]
tag = argv[0]
workload = argv[1] if len(argv) > 1 else "chair_4k"
note = argv[2] if len(argv) > 2 else ""
scene, W, H = WORKLOADS[workload]
G = os.path.join(ROOT, "gpurun_out")
KERNEL = "rm_render_jit"
PMC_LAUNCHES = 2      # the counter passes run --steps 64 --warmup 32: a warm-up launch of 32 frames, then the two timed ones, which are the ones counted

bench_line = open(os.path.join(G, tag + "_trace_bench.json")).read().strip().splitlines()[-1]
rec = json.loads(bench_line)
fpl = int(round(rec["config"].get("frames_per_launch", 1)))
lines = ["# %s: %s (%s.sce %dx%d), MI355X, one context, launches of %d frames back to back; all figures per launch%s" % (tag, workload, scene, W, H, fpl, (" -- " + note) if note else ""),
         "# command per pass: rocprofv3 [--kernel-trace --stats | --pmc <counters> --kernel-trace] -- python3 bench.py --workload %s --steps K --warmup W --no-cpu-baseline --no-single-frame" % workload,
         "# (tools/profile_round.sh, summarised by tools/profile_summary.py; every counter pass is a run of its own)", ""]

# ---- kernel trace
db_path = glob.glob(os.path.join(G, tag + "_trace", "**", "*.db"), recursive=True)
lines.append("## kernel trace (--steps 96 --warmup 16, %d frames per launch)" % fpl)
avg_us = None
if db_path:
    db = sqlite3.connect(db_path[0])
    lines.append("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
        lines.append("%-90s %8d %14.3f %12.3f %8.3f" % (name[:90], calls, total, avg, pct))
        if name.startswith(KERNEL):
            avg_us = avg
else:
    stats = glob.glob(os.path.join(G, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True)
    for row in csv.DictReader(open(stats[0])):
        lines.append("%-90s %8s %14.3f %12.3f %8s" % (row["Name"][:90], row["Calls"], float(row["TotalDurationNs"]) / 1e3,
                                                        float(row["AverageNs"]) / 1e3, row["Percentage"]))
        if row["Name"].startswith(KERNEL):
            avg_us = float(row["AverageNs"]) / 1e3
lines.append("")
lines += ["## bench line of the traced run", bench_line, ""]

# ---- counters: mean per launch over the timed launches
counters = {}
for path in sorted(glob.glob(os.path.join(G, tag + "_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(path)):
        if not r["Kernel_Name"].startswith(KERNEL):
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, by_dispatch in per.items():
        vals = [v for _, v in sorted(by_dispatch.items())][-PMC_LAUNCHES:]        # the timed launches are the last ones
        counters[name] = sum(vals) / len(vals)
lines.append("## counters, mean per launch of %s over the timed launches (sum over the chip)" % KERNEL)
for name, v in counters.items():
    lines.append("%-28s %14.0f" % (name, v))
lines.append("")

# ---- derived
rays = W * H * fpl                            # per launch
lines.append("## derived")
if "GRBM_GUI_ACTIVE" in counters:
    cyc = counters["GRBM_GUI_ACTIVE"] / 8
    lines.append("shader clock cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs)   %10.0f  (%.3f ms at 2.4 GHz)" % (cyc, cyc / 2.4e6))
    simd_cycles = cyc * 256 * 4
    if "SQ_INSTS_VALU" in counters:
        lines.append("wave-level VALU instructions per SIMD-cycle                  %10.3f  (v_add_f32 alone sustains 0.42: tools/microbench/valu_peak.hip)"
                     % (counters["SQ_INSTS_VALU"] / simd_cycles))
    if "SQ_ACTIVE_INST_VALU" in counters:
        lines.append("SQ_ACTIVE_INST_VALU per SIMD-cycle                           %10.3f" % (counters["SQ_ACTIVE_INST_VALU"] / simd_cycles))
if "SQ_INSTS_VALU" in counters:
    alg = rec.get("roofline", {}).get("algorithmic_flops_per_ray")
    lines.append("lane-level VALU instructions per ray                         %10.0f  (algorithmic FLOP per ray: %s)"
                 % (counters["SQ_INSTS_VALU"] * 64 / rays, "%.0f" % alg if alg else "not counted for this workload"))
    if "SQ_INSTS_SALU" in counters:
        lines.append("SALU / VALU instructions                                     %10.2f" % (counters["SQ_INSTS_SALU"] / counters["SQ_INSTS_VALU"]))
hbm = None
if "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
    # MI355X_MICROARCH.md: both in KiB-like units of 1 KB per count here (rocprofv3 derived metric, KB); FETCH_SIZE x2 on gfx950
    wr, rd = counters["WRITE_SIZE"] * 1024, counters["FETCH_SIZE"] * 1024 * 2
    hbm = wr + rd
    lines.append("HBM traffic per launch (%d frames): WRITE_SIZE %.1f MB + FETCH_SIZE x2 %.1f MB (gfx950 correction of the guide); algorithmic: %.1f MB of pixels"
                 % (fpl, wr / 1e6, rd / 1e6, rays * 4 / 1e6))
if avg_us is not None:
    lines.append("average kernel duration in the trace %.1f us; bench.py's own HIP-event average %.1f us" % (avg_us, rec["kernel_ms"] * 1e3))

os.makedirs(out_dir, exist_ok=True)
out_txt = os.path.join(out_dir, tag + ".txt")
with open(out_txt, "w") as f:
    f.write("\n".join(lines) + "\n")
with open(os.path.join(out_dir, tag + "_counters.json"), "w") as f:
    json.dump({"workload": workload, "frames_per_launch": fpl, "hbm_bytes_per_launch": hbm,
               "source": "profiles/%s.txt (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)" % tag,
               "counters": counters}, f, indent=1)
with open(os.path.join(out_dir, tag + "_bench.json"), "w") as f:
    f.write(bench_line + "\n")
print(open(out_txt).read())
