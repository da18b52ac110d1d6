#!/usr/bin/env python3
"""Development tool: rewrite the measured-kernel table of DESIGN.md §4 and the headline figures of DESIGN.md / README.md
from the files under profiles/ (r02_bench.json, r02_kernel_stats.md, r02_pmc_traffic.json), so that the prose never
drifts from the committed profile of the same run.  Usage: python tools/sync_docs.py [tag]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sp(x):
    v = int(round(x, -1))
    return "%d" % v if v < 1000 else "%d %03d" % (v // 1000, v % 1000)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    prof = os.path.join(ROOT, "profiles")
    ks = open(os.path.join(prof, tag + "_kernel_stats.md")).read()
    bench = json.load(open(os.path.join(prof, tag + "_bench.json")))
    traffic = json.load(open(os.path.join(prof, tag + "_pmc_traffic.json")))

    def avg(name):
        for line in ks.split("\n"):
            if name in line and line.startswith("| `"):
                return float(line.split("|")[4]) / 1e3
        raise SystemExit("kernel not in stats: " + name)

    def tr(name):
        for k, v in traffic.items():
            if isinstance(v, dict) and name in k:
                return v["hbm_bytes_per_launch"] / 1e9
        raise SystemExit("kernel not in traffic: " + name)

    bw = bench["roofline"]
    fw = bw["also"][0]
    agg_f, agg_b = "csr_aggregate_fwd_kernel<float, 4, 3, 3, false, false>", "csr_aggregate_bwd_kernel<float, 4, 3, 3, false, false>"
    tg1, tg2, tg3, tg0 = (avg("tallgemm_kernel<8, 8, 1>"), avg("tallgemm_kernel<4, 16, 2>"), avg("tallgemm_kernel<8, 8, 3>"),
                          avg("tallgemm_kernel<4, 16, 0>"))
    wa, wb = avg("linear_wgrad_kernel<8, 4, 2, 2, 2, true, false, true>"), avg("linear_wgrad_kernel<8, 2, 4, 2, 2, true, false, true>")
    lb, lf = avg("layernorm_act_bwd_kernel<float, 4, 5>"), avg("layernorm_act_fwd_kernel<float, 4, 5>")
    rows = {
        "| `csr_aggregate_fwd<float,4,RANK1,SOFTMAX>` (+ row maxima)":
            "| `csr_aggregate_fwd<float,4,RANK1,SOFTMAX>` (+ row maxima) | %.3f ms | %.3f ms | %s (%.2f) | %.2f GB | %.2f TB/s (%.2f) |" % (
                fw["avg_launch_ms"], avg(agg_f), sp(fw["achieved"]), fw["frac"], tr(agg_f[:-1]),
                fw["traffic"] / fw["avg_launch_ms"] / 1e9, fw["frac_hbm_counter"]),
        "| `tallgemm<8,8,LN-out>` / `<4,16,LN-in>`":
            "| `tallgemm<8,8,LN-out>` / `<4,16,LN-in>` (the fused MLP's 128→256 / 256→128) | — | %.3f / %.3f ms | — | 0.99 / 1.21 GB | — |" % (tg1, tg2),
        "| `tallgemm<8,8,LN-bwd>`":
            "| `tallgemm<8,8,LN-bwd>` (`dA = dY·W₂` → ReLU → LayerNorm backward, 128→256) | — | %.3f ms | %s (1.64 GB: %.2f) | 1.65 GB | — |" % (
                tg3, sp(1638.4 / tg3), 1638.4 / tg3 / 8000),
        "| `tallgemm<4,16>` plain":
            "| `tallgemm<4,16>` plain (input gradient of the first Linear, row maxima supplied) | — | %.3f ms | %s (0.98 GB: %.2f) | 0.99 GB | — |" % (
                tg0, sp(983.0 / tg0), 983.0 / tg0 / 8000),
        "| `linear_wgrad<8,·,fp16-split>`":
            "| `linear_wgrad<8,·,fp16-split>` (256×128 / 128×256 outputs; 0.262 / 0.285 ms with the exact bf16 split) | — | %.3f / %.3f ms | %s / %s (0.98 GB read: %.2f / %.2f) | 1.02 GB | — |" % (
                wa, wb, sp(983.0 / wa), sp(983.0 / wb), 983.0 / wa / 8000, 983.0 / wb / 8000),
        "| `layernorm_act_bwd<5>`":
            "| `layernorm_act_bwd<5>` (d=128, + identity-branch gradient) / `fwd<5>` (the d=256 backward now runs inside `tallgemm<8,8,LN-bwd>`: 0.406 ms before) | — | %.3f / %.3f ms | %s / %s (%.2f / %.2f) | 1.21 / 0.67 GB | — |" % (
                lb, lf, sp(1310.7 / lb), sp(655.4 / lf), 1310.7 / lb / 8000, 655.4 / lf / 8000),
    }
    bwd_tail = "| %.3f ms | %.3f + %.3f ms | %s (%.2f) | %.2f + %.2f GB | %.2f TB/s (%.2f) |" % (
        bw["avg_launch_ms"], avg(agg_b), avg("softmax_shift_kernel<float"), sp(bw["achieved"]), bw["frac"], tr(agg_b[:-1]),
        tr("softmax_shift"), bw["traffic"] / bw["avg_launch_ms"] / 1e9, bw["frac_hbm_counter"])
    path = os.path.join(ROOT, "DESIGN.md")
    out = []
    for line in open(path).read().split("\n"):
        for head, new in rows.items():
            if line.startswith(head):
                line = new
        if line.startswith("| `csr_aggregate_bwd<float,4,RANK1,SOFTMAX>` + `softmax_shift`"):
            line = re.sub(r"\| [0-9.]+ ms \| [0-9.]+ \+ [0-9.]+ ms \| .*$", bwd_tail, line)
        out.append(line)
    text = "\n".join(out)
    v, ms = bench["value"], bench["ms_per_step"]
    text = re.sub(r"`profiles/%s_bench.json`: [0-9 ]+ / [0-9.]+ ms;" % tag,
                  "`profiles/%s_bench.json`: %s / %.2f ms;" % (tag, ("%d %03d" % (int(v) // 1000, int(v) % 1000)), ms), text)
    open(path, "w").write(text)
    print("DESIGN.md synced: %.0f graphs/s, %.2f ms; fwd %.3f ms, bwd %.3f ms" % (v, ms, fw["avg_launch_ms"], bw["avg_launch_ms"]))


if __name__ == "__main__":
    main()
