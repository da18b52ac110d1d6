#!/usr/bin/env python3
"""dense_diff_pool at BASELINE configs[4] size (pooled graph of 4096 nodes, 1024 clusters, 256 channels, bf16):
the matrix-core product chain of csrc/diffpool_large.hip next to the library-GEMM formulation it replaces.

  python tools/bench_diffpool.py [--iters 20] [--json profiles/r02_diffpool_configs4.json]

FLOP accounting.  `reference_GFLOP` is the reference's own formulation (2KNC + 2KN^2 + 2K^2N + 2N^2K, the last
term being S S^T for the link loss); `executed_GFLOP` is what the chain here runs (the link loss via
||A||^2 - 2<S, A S> + ||S^T S||^2 replaces the 2N^2K of S S^T by the 2K^2N of S^T S).  MFMA utilisation is quoted on the
EXECUTED FLOP (the matrix cores did that much work); the speed-up is quoted on time."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))


def timed(fn, n):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=4096)
    ap.add_argument("--clusters", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--json", default=None)
    ap.add_argument("--skip-library", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="fp32: the three-term fp32-accurate form of the chain against the library's fp32 GEMMs")
    a = ap.parse_args()
    from mlgnn import dense
    N, K, C = a.nodes, a.clusters, a.channels
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    z = torch.randn(1, N, C, generator=g).to(dev).to(dt).requires_grad_(True)
    s = torch.randn(1, N, K, generator=g).to(dev).to(dt).requires_grad_(True)
    adj = (torch.rand(N, N, generator=g) + torch.eye(N)).to(dev).to(dt).unsqueeze(0)

    def fwd(fn):
        with torch.no_grad():
            return fn(z, adj, s)

    def fwd_bwd(fn):
        z.grad = s.grad = None
        x, aa, l, e = fn(z, adj, s)
        (x.float().sum() + aa.float().sum() + l.float() + e.float()).backward()

    # the operator alone: cotangents prepared once, no loss kernels between forward and backward
    cot = {}

    def fwd_bwd_op(fn):
        outs = fn(z, adj, s)
        if not cot:
            gen = torch.Generator(device=dev).manual_seed(11)
            cot["g"] = tuple(torch.randn(o.shape, generator=gen, device=dev, dtype=torch.float32).to(o.dtype) for o in outs)
        torch.autograd.grad(outs, (z, s), cot["g"])

    ref_flop = 2.0 * K * N * C + 2.0 * K * N * N + 2.0 * K * K * N + 2.0 * N * N * K
    exe_fwd = 2.0 * K * N * C + 2.0 * K * N * N + 2.0 * K * (2 * K) * N
    exe_bwd_sym = 2.0 * N * K * (C + 3 * K) + 2.0 * N * C * K
    exe_bwd = exe_bwd_sym + 2.0 * K * N * N
    if a.dtype == "fp32":                                   # three bf16 terms per product
        exe_fwd, exe_bwd, exe_bwd_sym = 3 * exe_fwd, 3 * exe_bwd, 3 * exe_bwd_sym
    res = {"config": "dense_diff_pool N=%d K=%d C=%d %s (BASELINE configs[4] DiffPool%s)" % (
               N, K, C, a.dtype, "" if a.dtype == "bf16" else "; fp32 inputs, three-term bf16 products"),
           "dense_peak_TFLOPs_bf16": 2500.0, "reference_fwd_GFLOP": ref_flop / 1e9, "executed_fwd_GFLOP": exe_fwd / 1e9,
           "executed_bwd_GFLOP": exe_bwd / 1e9}
    t_f = timed(lambda: fwd(dense.dense_diff_pool), a.iters)
    t_fb = timed(lambda: fwd_bwd(dense.dense_diff_pool), a.iters)
    t_fb_sym = timed(lambda: fwd_bwd(lambda *x: dense.dense_diff_pool(*x, adj_symmetric=True)), a.iters)
    t_op = timed(lambda: fwd_bwd_op(dense.dense_diff_pool), a.iters)
    t_op_sym = timed(lambda: fwd_bwd_op(lambda *x: dense.dense_diff_pool(*x, adj_symmetric=True)), a.iters)
    res["matrix_core_chain"] = {
        "fwd_ms": t_f * 1e3, "fwd_bwd_ms": t_fb * 1e3, "fwd_bwd_ms_adj_symmetric": t_fb_sym * 1e3,
        "fwd_TFLOPs_executed": exe_fwd / t_f / 1e12, "fwd_MFMA_utilisation_executed": exe_fwd / t_f / 2.5e15,
        "fwd_TFLOPs_reference_formulation": ref_flop / t_f / 1e12,
        "fwd_bwd_TFLOPs_executed": (exe_fwd + exe_bwd) / t_fb / 1e12,
        "fwd_bwd_MFMA_utilisation_executed": (exe_fwd + exe_bwd) / t_fb / 2.5e15,
        # forward + backward of the operator with prepared cotangents (fwd_bwd_ms above also times the harness's loss:
        # ~25 small ATen launches between the two)
        "fwd_bwd_op_ms": t_op * 1e3, "fwd_bwd_op_ms_adj_symmetric": t_op_sym * 1e3,
        "fwd_bwd_op_TFLOPs_executed": (exe_fwd + exe_bwd) / t_op / 1e12,
        "fwd_bwd_op_MFMA_utilisation_executed": (exe_fwd + exe_bwd) / t_op / 2.5e15}
    if not a.skip_library:
        l_f = timed(lambda: fwd(dense._diff_pool_library), a.iters)
        l_fb = timed(lambda: fwd_bwd(dense._diff_pool_library), a.iters)
        cot.clear()
        l_op = timed(lambda: fwd_bwd_op(dense._diff_pool_library), a.iters)
        res["library_gemm_formulation"] = {"fwd_ms": l_f * 1e3, "fwd_bwd_ms": l_fb * 1e3, "fwd_bwd_op_ms": l_op * 1e3,
                                           "fwd_TFLOPs": ref_flop / l_f / 1e12, "fwd_MFMA_utilisation": ref_flop / l_f / 2.5e15}
        res["speedup_fwd"] = l_f / t_f
        res["speedup_fwd_bwd"] = l_fb / t_fb
        res["speedup_fwd_bwd_op"] = l_op / t_op
    out = json.dumps(res, indent=1)
    print(out)
    if a.json:
        with open(a.json, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
