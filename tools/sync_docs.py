#!/usr/bin/env python3
"""Development tool: regenerate every number DESIGN.md / README.md quote from the files under profiles/ -- the text
between ``<!-- measured:TAG -->`` / ``<!-- /measured:TAG -->`` (the kernel table of the headline step) and
between ``<!-- headline:TAG -->`` / ``<!-- /headline:TAG -->`` (both files: step time, throughput, the other aggregators,
the DiffPool and stress figures) is REPLACED by what the profiles say, so the prose cannot drift from the committed
profile of the same run.  Usage: python tools/sync_docs.py [tag]      (default r03)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS = 640000                      # BASELINE configs[1]: 64 graphs x 10 000 nodes


def load(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    ks = open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.md")).read()
    bench = load(tag + "_bench.json")
    traffic = load(tag + "_pmc_traffic.json") or {}
    dense = load(tag + "_dense_kernels.json")
    dp = load(tag + "_diffpool_configs4.json")
    dp32 = load(tag + "_diffpool_configs4_fp32.json")
    stress = load(tag + "_stress_configs4_bf16.json")
    csr = load(tag + "_csr_build.json")

    def stat(name):
        """(calls, avg ms) of the first stats row whose kernel name contains `name`."""
        for line in ks.split("\n"):
            if line.startswith("| `") and name in line:
                f = line.split("|")
                return int(f[2]), float(f[4]) / 1e3
        return None

    def pmc(name):
        for k, v in traffic.items():
            if isinstance(v, dict) and name in k:
                return v["hbm_bytes_per_launch"] / 1e9
        return None

    rl = bench["roofline"]
    views = {rl["kernel"]: rl}
    for v in rl.get("also", []):
        views[v["kernel"]] = v
    lines = ["| kernel (3 launches of each per step) | avg launch (rocprofv3) | algorithmic bytes | GB/s on them (of 8 TB/s) | PMC HBM bytes "
             "(`2·FETCH_SIZE + WRITE_SIZE`) | TB/s on those |", "|---|---|---|---|---|---|"]
    gb = lambda floats: floats * 4.0 / 1e9                               # noqa: E731
    table = [
        ("csr_aggregate_fwd_kernel<float, 4, 3, 3, false, false>", "`csr_aggregate_fwd<float,4,RANK1,SOFTMAX>` (+ row maxima, lse)",
         views.get("csr_aggregate_fwd/softmax/rank1", {}).get("algorithmic_bytes_per_launch", 0) / 1e9),
        ("csr_aggregate_bwd_kernel<float, 4, 3, 3, false, false>", "`csr_aggregate_bwd<float,4,RANK1,SOFTMAX>` (rescaled cotangent from the producing GEMM)",
         views.get("csr_aggregate_bwd/softmax/rank1", {}).get("algorithmic_bytes_per_launch", 0) / 1e9),
        ("linear_bwd_kernel<128, 256, 0>", "`linear_bwd<128,256,LN>`: dW₂, db₂, `dA = go·W₂` → ReLU → LayerNorm backward, one pass",
         gb(N_ROWS * (128 + 2 * 256))),
        ("linear_bwd_kernel<256, 128, 2>", "`linear_bwd<256,128,SHIFT>`: dW₁, db₁, `gx = gh·W₁`, `gx·2^(−lse)`, one pass",
         gb(N_ROWS * (256 + 4 * 128))),
        ("tallgemm_kernel<8, 8, 1, false, false>", "`tallgemm<8,8,LN-out>` (128→256, result layer-normalised)", gb(N_ROWS * (128 + 256))),
        ("tallgemm_kernel<4, 16, 2, true, false>", "`tallgemm<4,16,LN-in,POST>` (256→128 + residual + the next block's norm + ReLU)",
         gb(N_ROWS * (256 + 3 * 128))),
        ("layernorm_act_bwd_kernel<float, 4, 5>", "`layernorm_act_bwd<5>` (d=128, + identity-branch gradient)", gb(N_ROWS * 4 * 128)),
    ]
    for key, label, alg in table:
        st = stat(key)
        if st is None:
            continue
        calls, ms = st
        hb = pmc(key.split("<")[0] + "<" + key.split("<")[1]) if "<" in key else pmc(key)
        lines.append("| %s | %.3f ms | %.2f GB | %.0f (%.2f) | %s | %s |" % (
            label, ms, alg, alg / ms * 1e3, alg / ms * 1e3 / 8000.0,
            "%.2f GB" % hb if hb else "—", "%.2f (%.2f)" % (hb / ms, hb / ms / 8.0) if hb else "—"))
    measured = "\n".join(lines)

    also = {a["aggr"]: a for a in bench.get("also_aggr", [])}
    head = ["**Round 3, `profiles/%s_bench.json`: %.0f graphs/s = %.2f ms per 64-graph step** (softmax; `bench.py` defaults)"
            % (tag, bench["value"], bench["ms_per_step"])]
    if also:
        head.append("; " + ", ".join("%s %.2f ms (%.0f graphs/s)" % (k, v["ms_per_step"], v["value"]) for k, v in sorted(also.items())))
    if "no_overlap_ms_per_step" in bench:
        head.append("; topology built in line (`--no-overlap`): %.2f ms" % bench["no_overlap_ms_per_step"])
    head.append(".  Aggregation (HIP events in the timed region): forward %.3f ms, backward %.3f ms per launch of 64 graphs"
                % (views["csr_aggregate_fwd/softmax/rank1"]["avg_launch_ms"], views["csr_aggregate_bwd/softmax/rank1"]["avg_launch_ms"]))
    cb = bench.get("cpu_baseline")
    if cb:
        head.append("; CPU oracle on the same box: %.2f graphs/s on %d cores" % (cb["value"], cb["cores"]))
    head.append(".")
    if dense:
        k = {r["kernel"].split()[0]: r for r in dense["kernels"]}
        if "F2" in k and "W2" in k:
            head.append("  One GENConv layer's dense kernels on their own (`profiles/%s_dense_kernels.json`, operands streamed from "
                        "HBM): one-pass Linear backward %.3f + %.3f ms against %.3f + %.3f + %.3f + %.3f ms for the two-kernel form."
                        % (tag, k["F2"]["ms"], k["F1"]["ms"], k["W2"]["ms"], k["B2"]["ms"], k["W1"]["ms"], k["B1"]["ms"]))
    if dp:
        m = dp["matrix_core_chain"]
        head.append("  configs[4] DiffPool 4096 / 1024 / 256 bf16 (`profiles/%s_diffpool_configs4.json`): forward %.3f ms (%.0f %% of the "
                    "dense bf16 peak on the executed FLOP), forward + backward %.3f ms (operator alone %.3f ms, %.3f with a symmetric "
                    "adjacency)" % (tag, m["fwd_ms"], 100 * m["fwd_MFMA_utilisation_executed"], m["fwd_bwd_ms"], m["fwd_bwd_op_ms"],
                                    m["fwd_bwd_op_ms_adj_symmetric"]))
        if dp32:
            m32 = dp32["matrix_core_chain"]
            head.append("; fp32 inputs (three-term products): %.3f / %.3f ms" % (m32["fwd_ms"], m32["fwd_bwd_ms"]))
        head.append(".")
    if stress:
        head.append("  configs[4] stress step (bf16, 28 layers, `profiles/%s_stress_configs4_bf16.json`): %.1f ms." % (tag, stress["step_ms"]))
    if csr:
        head.append("  Topology build of a configs[1] batch on its own (`profiles/%s_csr_build.json`): %.2f ms."
                    % (tag, csr["csr_build_plus_edge_table_ms"]))
    headline = "".join(head)

    for fname in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, fname)
        text = open(path).read()
        for marker, body in (("measured", measured), ("headline", headline)):
            pat = re.compile(r"(<!-- %s:%s -->\n).*?(\n<!-- /%s:%s -->)" % (marker, tag, marker, tag), re.S)
            text, n = pat.subn(lambda mm: mm.group(1) + body + mm.group(2), text)
            if n:
                print("%s: %s block replaced (%d)" % (fname, marker, n))
        open(path, "w").write(text)


if __name__ == "__main__":
    main()
