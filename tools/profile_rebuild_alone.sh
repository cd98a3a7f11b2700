#!/bin/bash
# usage (GPU box, repo root): tools/profile_rebuild_alone.sh TAG
# rm::assemble_luminance_kernel alone on the chip (tools/rebuild_rate.py: launches of 32 frames, 8 ranks' bands), per
# image size a kernel trace and the two HBM counter passes, each a rocprofv3 run of its own; printed as a profiles/ file
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
for size in "3840 2160" "7680 4320"; do
    name=${size/ /x}
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_${name}_trace -o t -- python3 $root/tools/rebuild_rate.py $size > $root/gpurun_out/${tag}_${name}_print.txt
    for pass in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_${name}_pmc_$pass -o pmc -- python3 $root/tools/rebuild_rate.py $size > /dev/null
    done
done
cd $root && python3 - "$tag" <<'PY'
import csv, glob, os, sqlite3, sys
tag = sys.argv[1]
G = "gpurun_out"
out = ["# %s: rm::assemble_luminance_kernel alone on one MI355X (tools/rebuild_rate.py W H: launches of 32 frames whose bands 8 ranks hold," % tag,
       "# an even deal and a 3:5 deal, 6 launches each); rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE, each a run of its own",
       "# (tools/profile_rebuild_alone.sh).  5 B/px algorithmic: 1 B/px of luminance read, 4 B/px of RGBA8 written.", ""]
for name, (w, h) in (("3840x2160", (3840, 2160)), ("7680x4320", (7680, 4320))):
    db = sqlite3.connect(glob.glob(os.path.join(G, "%s_%s_trace" % (tag, name), "**", "*.db"), recursive=True)[0])
    out.append("## %s" % name)
    out.append("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    avg = None
    for kname, calls, total, a, pct in db.execute("select * from top_kernels"):
        out.append("%-70s %8d %14.3f %12.3f %8.3f" % (kname[:70], calls, total, a, pct))
        if kname.startswith("rm::assemble_luminance_kernel"):
            avg = a
    cnt = {}
    for path in glob.glob(os.path.join(G, "%s_%s_pmc_*" % (tag, name), "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"].startswith("rm::assemble_luminance_kernel"):
                per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for k, v in per.items():
            cnt[k] = sum(v.values()) / len(v)
    px = w * h * 32
    out.append("per launch of 32 frames: %.1f us = %.2f us per frame; 5 B/px over that: %.0f GB/s = %.3f of the 8000 GB/s HBM peak"
               % (avg, avg / 32, 5 * px / avg / 1e3, 5 * px / avg / 1e3 / 8000))
    out.append("WRITE_SIZE %.1f MB per launch = %.2f B/px (16-byte stores: exact);  FETCH_SIZE %.1f MB as read = %.2f B/px (4-byte loads per lane: a width the guide does not calibrate)"
               % (cnt["WRITE_SIZE"] * 1024 / 1e6, cnt["WRITE_SIZE"] * 1024 / px, cnt["FETCH_SIZE"] * 1024 / 1e6, cnt["FETCH_SIZE"] * 1024 / px))
    out.append("the script's own HIP-event figures of the traced run:")
    out += ["  " + l.strip() for l in open(os.path.join(G, "%s_%s_print.txt" % (tag, name))) if "us per frame" in l]
    out.append("")
os.makedirs(os.path.join(G, "profiles_" + tag), exist_ok=True)
open(os.path.join(G, "profiles_" + tag, tag + ".txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
