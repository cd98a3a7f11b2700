#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv values per kernel and counter (last dispatch of each kernel)."""
import csv
import sys
from collections import OrderedDict

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    last = OrderedDict()
    for r in rows:
        key = (r["Kernel_Name"][:40], r["Counter_Name"])
        last[key] = (r["Dispatch_Id"], float(r["Counter_Value"]))
    for (k, c), (d, v) in last.items():
        print("%-42s %-28s dispatch %-4s %18.0f" % (k, c, d, v))
