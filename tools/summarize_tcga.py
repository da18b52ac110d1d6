#!/usr/bin/env python3
"""profiles/<tag>_tcga.json from the passes of tools/profile_round.sh over the model the reference ships configs for
(MultilevelGNN, gnn_name sage, config/kirc.yaml and gbm.yaml shape): ms per step, share of the kernel time spent in
hand-written (mlgnn::) kernels, and -- for the aggregation and projection kernels -- algorithmic, counter-measured and
compulsory bytes per launch against the 8 TB/s HBM peak.

    python tools/summarize_tcga.py <tag> [commit]
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0            # GB/s
NN, E1, G, S = 5135 * 3, 60000, 25015, 438
SHAPES = {"kirc": dict(B=64, dims=(32, 64, 32), k=3), "gbm": dict(B=32, dims=(64, 64, 32), k=2)}


def _own(name):
    """a kernel of libmlgnn.so (rocprofv3 leaves some names mangled: _ZN5mlgnn...)"""
    return "mlgnn::" in name or "_ZN5mlgnn" in name


def main():
    tag = sys.argv[1]
    commit = sys.argv[2] if len(sys.argv) > 2 else "unknown"
    out = {"_source": {"tag": tag, "commit": commit, "tool": "tools/profile_round.sh -> tools/summarize_tcga.py",
                       "peak_GBps": PEAK,
                       "bytes": "algorithmic = SURVEY 8(d) edge-gather formulation (one gathered row per edge, no cache "
                                "credit); counter = 2*FETCH_SIZE + WRITE_SIZE (separate --pmc passes, MI355X_MICROARCH.md); "
                                "compulsory = every row read once / written once + the index arrays"}}
    for shape, cfg in SHAPES.items():
        ent = {}
        for key, name in (("ms_per_step", "tcga_%s_%s.json" % (shape, tag)),
                          ("ms_per_step_without_shared_topology", "tcga_%s_noshare_%s.json" % (shape, tag))):
            p = os.path.join(ROOT, "gpurun_out", name)
            if os.path.exists(p):
                d = json.load(open(p))
                ent[key] = d["ms_per_step"]
                ent.setdefault("workload", d["workload"])
                ent.setdefault("batch", d["batch"])
        hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_tcga_%s_%s" % (shape, tag), "**", "*kernel_stats.csv"),
                                recursive=True))
        if not hits:
            continue
        rows = list(csv.DictReader(open(hits[-1])))
        steps = 13                                        # 3 warm-up + 10 timed
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        own = sum(float(r["TotalDurationNs"]) for r in rows if _own(r["Name"]))
        ent["kernel_ms_per_step"] = total / steps / 1e6
        ent["mlgnn_share_of_kernel_time"] = own / total
        ent["launches_per_step"] = sum(int(r["Calls"]) for r in rows) / steps
        ent["mlgnn_launches_per_step"] = sum(int(r["Calls"]) for r in rows if _own(r["Name"])) / steps
        ent["top_non_mlgnn"] = [{"kernel": r["Name"][:90], "us_per_step": float(r["TotalDurationNs"]) / steps / 1e3}
                                for r in rows if not _own(r["Name"])][:6]
        tr_path = os.path.join(ROOT, "profiles", "%s_tcga_%s_pmc_traffic.json" % (tag, shape))
        traffic = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
        B, k = cfg["B"], cfg["k"]
        N, Eb = B * NN, B * (E1 + NN)                      # rows; edges incl. the added self loops
        kernels = []
        for r in rows:
            name = r["Name"]
            avg_us = float(r["AverageNs"]) / 1e3
            m = re.search(r"mlgnn::csr_short_(fwd|bwd)_kernel<(\d+)", name)
            alg = comp = None
            if m:
                d = 4 * int(m.group(2))
                # gathered rows + col + weight + rowptr + output rows
                alg = Eb * d * 4 + Eb * 8 + (N + 1) * 4 + N * d * 4
                comp = 2 * N * d * 4 + Eb * 8 + (N + 1) * 4
            m2 = re.search(r"mlgnn::segment_project_(fwd|bwd_x|bwd_w)_kernel<float, 4, (\d)>", name)
            if m2:
                C = cfg["dims"][2]
                gather = B * G * C * 4
                tables = B * G * 8 + G * k * 4
                small = B * S * k * C * 4
                if m2.group(1) == "fwd":                    # B G C s + B G 8 + G k 4 + B C 438 k s
                    alg, comp = gather + tables + small, N * C * 4 + tables + small
                elif m2.group(1) == "bwd_x":                # K cotangent rows per member + the grad_x rows written
                    alg, comp = B * G * k * C * 4 + tables + N * C * 4, small + tables + N * C * 4
                else:                                       # x rows gathered per member + the cotangent rows + [B G, k] partials
                    alg, comp = gather + tables + small + B * G * k * 4, N * C * 4 + tables + small + B * G * k * 4
            if alg is None:
                continue
            key = name.split("(")[0].replace("void ", "")
            cnt = (traffic.get(key) or {}).get("hbm_bytes_per_launch")
            secs = avg_us * 1e-6
            kernels.append({"kernel": key, "launches_per_step": int(r["Calls"]) / steps, "avg_us": avg_us,
                            "algorithmic_bytes": alg, "frac_algorithmic": alg / secs / 1e9 / PEAK,
                            "counter_bytes": cnt, "frac_counter": (cnt / secs / 1e9 / PEAK) if cnt else None,
                            "compulsory_bytes": comp, "frac_compulsory": comp / secs / 1e9 / PEAK})
        ent["kernels"] = kernels
        out[shape] = ent
    path = os.path.join(ROOT, "profiles", "%s_tcga.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {s: {k: round(v, 3) for k, v in e.items() if isinstance(v, float)} for s, e in out.items() if s != "_source"})


if __name__ == "__main__":
    main()
