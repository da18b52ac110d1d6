#!/usr/bin/env python3
"""Timeline summary of the LAST training step in a rocprofv3 ``--kernel-trace`` CSV: per queue busy time, the union of
busy intervals, idle gaps on the busiest queue, and the largest gaps with the kernels around them.  Development tool.
    python tools/trace_gaps.py "gpurun_out/prof/**/*kernel_trace.csv" [n_steps_back]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    hits = sorted(glob.glob(sys.argv[1], recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(hits[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam_step" in r["Kernel_Name"] or "adam_kernel" in r["Kernel_Name"]]
    if len(marks) < 3:
        raise SystemExit("fewer than 3 optimizer steps in the trace")
    lo, hi = marks[-3] + 1, marks[-2] + 1          # one whole step between two optimizer launches (not the last: extras follow)
    step = rows[lo:hi]
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    per_q = defaultdict(list)
    for r in step:
        per_q[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    print("step: %d dispatches, span %.3f ms" % (len(step), (t1 - t0) / 1e6))
    iv = sorted((s, e) for q in per_q.values() for s, e, _ in q)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print("union of busy intervals %.3f ms (idle %.3f ms)" % (union / 1e6, (t1 - t0 - union) / 1e6))
    for q, ks in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        print("queue %s: %d dispatches, busy %.3f ms" % (q, len(ks), busy / 1e6))
    main_q = max(per_q.values(), key=lambda ks: sum(e - s for s, e, _ in ks))
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(main_q, main_q[1:]):
        if s1 > e0:
            gaps.append((s1 - e0, n0, n1))
    print("busiest queue: %d gaps, total %.3f ms; gaps > 5 us: %d totalling %.3f ms" % (
        len(gaps), sum(g for g, _, _ in gaps) / 1e6, sum(1 for g, _, _ in gaps if g > 5000),
        sum(g for g, _, _ in gaps if g > 5000) / 1e6))
    for g, a, b in sorted(gaps, reverse=True)[:15]:
        print("  %7.1f us between %s -> %s" % (g / 1e3, a[:60], b[:60]))


if __name__ == "__main__":
    main()
