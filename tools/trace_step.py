#!/usr/bin/env python3
"""Ordered kernel list of the LAST training step from a rocprofv3 ``--kernel-trace`` CSV
(``*_kernel_trace.csv``): one line per dispatch with its duration and the idle gap before it.
Steps are delimited by the first kernel of the device CSR build.  Development tool."""
import csv
import glob
import os
import sys


def main():
    pat = sys.argv[1]
    hits = sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(hits[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "csr_prepare_kernel" in r["Kernel_Name"]]
    # one CSR build has several launches of its first kernel close together: keep marks > 1000 dispatches apart
    starts = [m for j, m in enumerate(marks) if j == 0 or m - marks[j - 1] > 200]
    lo = starts[-1] if starts else 0
    prev_end = None
    total = busy = 0
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev_end is None else max(0, s - prev_end)
        name = r["Kernel_Name"].replace("void ", "")[:100]
        print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, name))
        busy += e - s
        prev_end = max(prev_end or e, e)
    total = prev_end - t0
    print("# last step: %d dispatches, span %.2f ms, busy %.2f ms" % (len(rows) - lo, total / 1e6, busy / 1e6))


if __name__ == "__main__":
    main()
