#!/usr/bin/env python3
"""pmc_summary.py TAG: counters of the LAST dispatch of every kernel in gpurun_out/TAG_pmc_*/ (rocprofv3 --pmc
csv output, summed over the chip's XCDs and shader engines)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(path)):
        per.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        per[(r["Kernel_Name"][:40], r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for key, by in per.items():
        out[key] = by[max(by)]
for (k, c), v in sorted(out.items()):
    print("%-42s %-28s %18.0f" % (k, c, v))
