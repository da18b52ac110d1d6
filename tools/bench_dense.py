#!/usr/bin/env python3
"""Development tool: every dense kernel of one GENConv layer (MLP 128 -> 256 -> 128 over 640 000 node rows, BASELINE
configs[1]) timed on its own with HIP events; bytes = what the kernel must read + write once."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn import dense as D  # noqa: E402
from mlgnn import norm as NM  # noqa: E402


def timed(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=640000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = "cuda:0"
    N, d, h = a.rows, a.d, 2 * a.d
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)           # noqa: E731
    x, res, go = r(N, d), r(N, d), r(N, d)
    w1, b1, w2, b2 = r(h, d) * 0.1, r(h) * 0.1, r(d, h) * 0.1, r(d) * 0.1
    gam, bet = torch.rand(h, device=dev) + 0.5, r(h) * 0.2
    pg, pb = torch.rand(d, device=dev) + 0.5, r(d) * 0.2
    xmax, gomax = x.abs().amax(1), go.abs().amax(1)
    xhat, rstd, amax = D.tall_matmul_nt(x, w1, b1, None, xmax, ln=("out", gam, bet, 1e-5))
    out, y, mean2, rstd2 = D.tall_matmul_lnin_postln(xhat, w2, b2, res, amax, gam, bet, (pg, pb, 1e-5, True))
    gh, _, _, ghmax = D.tall_matmul_ln_backward(go, w2, xhat, rstd, gam, bet, gomax)
    MB = 1e6
    rows = []

    def add(name, fn, nbytes):
        ms = timed(fn, a.iters)
        rows.append(dict(kernel=name, ms=ms, GB=nbytes / 1e9, TBps=nbytes / ms / 1e9))
        print("%-46s %7.3f ms  %6.2f GB  %5.2f TB/s  (%.2f of 8)" % (name, ms, nbytes / 1e9, nbytes / ms / 1e9,
                                                                      nbytes / ms / 1e9 / 8.0), flush=True)

    add("G1  tallgemm<LN-out> x[N,%d] -> xhat[N,%d]" % (d, h),
        lambda: D.tall_matmul_nt(x, w1, b1, None, xmax, ln=("out", gam, bet, 1e-5)), N * (d + h) * 4)
    add("G2  tallgemm<LN-in> xhat -> out (+res)", lambda: D.tall_matmul_nt(xhat, w2, b2, res, amax, ln=("in", gam, bet)),
        N * (h + 2 * d) * 4)
    add("G2p tallgemm<LN-in, POST> xhat -> out, y (+res)",
        lambda: D.tall_matmul_lnin_postln(xhat, w2, b2, res, amax, gam, bet, (pg, pb, 1e-5, True)), N * (h + 3 * d) * 4)
    add("LNf layernorm_act_fwd out -> y", lambda: NM.layer_norm_act(out, pg, pb, 1e-5, True), N * 2 * d * 4)
    add("LNb layernorm_act_bwd (+identity)", lambda: NM.ln_backward_saved(go, out, pg, pb, mean2, rstd2, True, extra=res),
        N * 4 * d * 4)
    add("W2  linear_wgrad go^T act(xhat)", lambda: D._wgrad(go, xhat, gam, bet, go_max=gomax, x_max=amax), N * (d + h) * 4)
    add("B2  tallgemm<LN-bwd> go -> gh", lambda: D.tall_matmul_ln_backward(go, w2, xhat, rstd, gam, bet, gomax),
        N * (d + 2 * h) * 4)
    add("W1  linear_wgrad gh^T x", lambda: D._wgrad(gh, x, go_max=ghmax, x_max=xmax), N * (d + h) * 4)
    add("B1  tallgemm gh -> gx", lambda: D.tall_matmul_nt(gh, w1, row_max=ghmax, bt_transposed=True), N * (d + h) * 4)
    if D.linear_backward_supported(N, d, h, D.LB_LN):
        lse = r(N, d)
        add("F2  linear_bwd<LN>: go, xhat -> gh, dW2 (one pass)",
            lambda: D.linear_backward(go, w2, xhat, gomax, amax, D.LB_LN, rstd=rstd, gamma=gam, beta=bet), N * (d + 2 * h) * 4)
        add("F1  linear_bwd<shift>: gh, x -> gx, gt, dW1 (one pass)",
            lambda: D.linear_backward(gh, w1, x, ghmax, xmax, D.LB_SHIFT, lse=lse), N * (h + 4 * d) * 4)
        add("F1p linear_bwd<plain>: gh, x -> gx, dW1 (one pass)",
            lambda: D.linear_backward(gh, w1, x, ghmax, xmax, D.LB_PLAIN), N * (h + 2 * d) * 4)
    tot = sum(r_["ms"] for r_ in rows if not r_["kernel"].startswith(("G2 ", "LNf", "F")))
    print("layer forward + backward with the POST epilogue (G1 G2p | LNb W2 B2 W1 B1): %.3f ms" % tot)
    if a.json:
        json.dump(dict(rows=N, d=d, kernels=rows, layer_ms=tot), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
