#!/usr/bin/env python3
"""Print the per-kernel summary (calls, total us, average us, % of GPU time) of a
rocprofv3 --kernel-trace --stats run from its rocpd sqlite database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    print("%-90s %8d %14.3f %12.3f %8.3f" % (name[:90], calls, total, avg, pct))
