#!/usr/bin/env python3
"""Print per-kernel call counts and average durations from a rocprofv3 ``*_kernel_stats.csv``
(``--kernel-trace --stats``), optionally filtered by a substring.  Development tool."""
import csv
import glob
import os
import sys

hits = sorted(glob.glob(sys.argv[1], recursive=True), key=os.path.getmtime)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(hits[-1])):
    if pat in r["Name"]:
        print("%6s calls  %9.1f us avg  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
