#!/usr/bin/env python3
"""Condense rocprofv3 output (``--kernel-trace --stats`` CSVs, optional ``--pmc`` CSVs) into the
small files kept under ``profiles/``.

  python tools/summarize_prof.py --stats gpurun_out/prof/**/_kernel_stats.csv --tag r01 \
         [--pmc-fetch <counter_collection.csv>] [--pmc-write <counter_collection.csv>]
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys_path = os.path.join(ROOT, "multilevel-gnn_amd")
import sys  # noqa: E402
if sys_path not in sys.path:
    sys.path.insert(0, sys_path)


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("no file matches %s" % pattern)
    return hits[-1]


def pmc_per_kernel(path, counter):
    """-> {kernel short name: mean counter value per dispatch} for mlgnn kernels."""
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "mlgnn::" not in name:
            continue
        acc[name.split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def _own(name):
    """a kernel of libmlgnn.so (rocprofv3 leaves some names mangled: _ZN5mlgnn...)"""
    return "mlgnn::" in name or "_ZN5mlgnn" in name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--cmd", default="")
    ap.add_argument("--pmc-cmd", default="")
    ap.add_argument("--commit", default="")
    ap.add_argument("--pmc-fetch")
    ap.add_argument("--pmc-write")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--no-traffic-json", action="store_true",
                    help="write only profiles/<tag>_pmc_traffic.json (bench.py's profiles/traffic.json belongs to the headline run)")
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    rows = list(csv.DictReader(open(find(a.stats))))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = os.path.join(ROOT, "profiles", "%s_kernel_stats.md" % a.tag)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n\n" % a.tag)
        if a.cmd:
            f.write("Command: `%s`\n\n" % a.cmd)
        f.write("Total kernel time: %.2f ms over %d distinct kernels.\n\n" % (total / 1e6, len(rows)))
        own = sum(float(r["TotalDurationNs"]) for r in rows if _own(r["Name"]))
        own_calls = sum(int(r["Calls"]) for r in rows if _own(r["Name"]))
        calls = sum(int(r["Calls"]) for r in rows)
        f.write("Hand-written (`mlgnn::`) kernels: %.1f %% of kernel time, %d of %d launches.\n\n"
                % (100.0 * own / max(total, 1.0), own_calls, calls))
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:a.top]:
            f.write("| `%s` | %s | %.2f | %.1f | %.1f | %.1f | %.1f |\n" % (
                r["Name"][:110].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
        f.write("\n## hand-written kernels (libmlgnn.so)\n\n| kernel | calls | avg us |\n|---|---|---|\n")
        for r in rows:
            if _own(r["Name"]):
                f.write("| `%s` | %s | %.1f |\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
    print("wrote", out)
    if a.pmc_fetch and a.pmc_write:
        fetch, nf = pmc_per_kernel(find(a.pmc_fetch), "FETCH_SIZE")
        write, _ = pmc_per_kernel(find(a.pmc_write), "WRITE_SIZE")
        traffic, detail, most = {}, {}, {}
        for k in fetch:
            # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
            # exactly half of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact.
            hbm = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
            short = k.split("mlgnn::")[1].split("<")[0].replace("_kernel", "")
            detail[k] = {"dispatches": nf[k], "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0),
                         "hbm_bytes_per_launch": hbm}
            # several instantiations of one kernel may have run (bench.py also times the max / mean aggregators): the
            # entry bench.py reads for the HEADLINE kernel is the instantiation launched most often (softmax: 3 layers x
            # every step of the timed run and of both side runs' warm-up), not the one that moved the most bytes
            if nf[k] > most.get(short, -1) or (nf[k] == most.get(short) and hbm > traffic.get(short, 0.0)):
                most[short] = nf[k]
                traffic[short] = hbm
        # per aggregator ("csr_aggregate_fwd/max", ...: bench.py's also_aggr legs): template arguments <T, VEC, MODE, AGGR,
        # ...> with AGGR 0 = sum / mean (the bench only runs mean), 2 = max, 3 = softmax; the main launch plus the
        # long-row (VIRT) launch of the same call
        import re
        names = {0: "mean", 2: "max", 3: "softmax"}
        per_aggr = {}
        for k, d in detail.items():
            m = re.search(r"mlgnn::(csr_aggregate_(?:fwd|bwd))_kernel<\w+, \d+, \d+, (\d+),", k)
            if m and int(m.group(2)) in names:
                key = "%s/%s" % (m.group(1), names[int(m.group(2))])
                per_aggr[key] = per_aggr.get(key, 0.0) + d["hbm_bytes_per_launch"]
        traffic.update(per_aggr)
        # one mlgnn_csr_aggregate_bwd call = the softmax shift pre-pass + the main kernel: bench.py times the call,
        # so its traffic entry is the sum of the two launches
        if "softmax_shift" in traffic and "csr_aggregate_bwd" in traffic:
            traffic["csr_aggregate_bwd"] += traffic["softmax_shift"]
            if "csr_aggregate_bwd/softmax" in traffic:
                traffic["csr_aggregate_bwd/softmax"] += traffic["softmax_shift"]
        # provenance travels with the numbers: bench.py refuses them when the kernel sources have changed since
        import build_native
        blob = {"_source": {"tag": a.tag, "commit": a.commit or os.environ.get("MLGNN_COMMIT", "unknown"),
                            "command": a.pmc_cmd or a.cmd, "kernel_sources_sha256": build_native.sources_digest(),
                            "workload": "64x10000x160000x128",       # graphs per GPU x nodes x edges x hidden (bench.py defaults)
                            "counters": "2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes "
                                        "(MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of a 16 B/lane stream on gfx950)"}}
        blob.update(traffic)
        if not a.no_traffic_json:
            json.dump(blob, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
        detail["_source"] = blob["_source"]
        json.dump(detail, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % a.tag), "w"), indent=1)
        print("wrote traffic", traffic)


if __name__ == "__main__":
    main()
