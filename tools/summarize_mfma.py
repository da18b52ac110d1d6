#!/usr/bin/env python3
"""MFMA counters of the large GEMM (csrc/gemm_nt.hip) from a rocprofv3 --pmc pass:
  python tools/summarize_mfma.py <counter_collection.csv glob> <kernel_stats.csv glob> <out.json> [commit]
Per distinct product (grouped by its MFMA instruction count): SQ_VALU_MFMA_BUSY_CYCLES (= 16 cycles per
v_mfma_f32_16x16x32_bf16, i.e. 1024 FLOP per busy cycle, summed over the 1024 SIMDs), SQ_INSTS_VALU_MFMA_MOPS_BF16, GRBM_GUI_ACTIVE (sum over the 8
XCDs) and the utilisation  busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import json
import sys


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("no file matches %s" % pattern)
    return hits[-1]


def main():
    pmc, stats, out = find(sys.argv[1]), find(sys.argv[2]), sys.argv[3]
    commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(pmc)):
        if "gemm_nt" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    groups = collections.defaultdict(list)
    for v in d.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            groups[v["SQ_VALU_MFMA_BUSY_CYCLES"]].append(v)
    rows = []
    for busy, vs in sorted(groups.items()):
        n = len(vs)
        grbm = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in vs) / n
        rows.append({"mfma_instructions": busy / 16.0, "GFLOP": busy / 16.0 * 16 * 16 * 32 * 2 / 1e9, "dispatches": n,
                     "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_VALU_MFMA_MOPS_BF16": vs[0].get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
                     "SQ_BUSY_CYCLES": sum(v.get("SQ_BUSY_CYCLES", 0.0) for v in vs) / n, "GRBM_GUI_ACTIVE": grbm,
                     "mfma_utilisation": busy / 1024.0 / (grbm / 8.0) if grbm else None})
    gemm = [r for r in csv.DictReader(open(stats)) if "gemm_nt" in r["Name"]]
    blob = {"_source": {"commit": commit, "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                        "SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -- python3 tools/bench_diffpool.py "
                        "--skip-library --iters 10",
                        "note": "mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
                                "GRBM_GUI_ACTIVE is summed over the 8 XCDs and reads high on dispatches this short "
                                "(MI355X_MICROARCH.md, DVFS give-back), so the figure is a lower bound on the busy "
                                "fraction at the clock the chip actually held"},
            "products": rows,
            "gemm_nt_kernel_stats": [{"calls": r["Calls"], "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                      "max_us": float(r["MaxNs"]) / 1e3} for r in gemm]}
    json.dump(blob, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
