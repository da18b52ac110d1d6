#!/usr/bin/env python3
"""BASELINE configs[4] stress run: ONE synthetic graph N=200 000 nodes / E=3 000 000 edges (ER, seed 7),
d=256, 28-layer DeeperGCN (res+, LayerNorm, softmax aggregation), then dense_diff_pool on a 4096-node
pooled graph with 1024 clusters.  ``--dtype bf16`` is the config's named storage type (bf16 activations / weights,
fp32 accumulation inside every kernel); fp32 is the default.  Prints per-kernel algorithmic GB/s from HIP-event
timing and the step time; ``--host-profile`` adds host-side profiles of one step.

  python tools/stress.py [--layers 28] [--hidden 256] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200000)
    ap.add_argument("--edges", type=int, default=3000000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--host-profile", action="store_true", help="cProfile one step (host-side time by function)")
    a = ap.parse_args()
    from _util import make_args
    from mlgnn import ops
    from mlgnn.dense import dense_diff_pool
    from models import get_model
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    ei = torch.randint(0, a.nodes, (2, a.edges), generator=gen)
    batch = SimpleNamespace(x=torch.randn(a.nodes, 3, generator=gen).to(dev), edge_index=ei.to(dev),
                            edge_attr=torch.rand(a.edges, 1, generator=gen).to(dev),
                            batch=torch.zeros(a.nodes, dtype=torch.long, device=dev), age=torch.zeros(1, device=dev),
                            pathway_node_attr=None, node_size=torch.tensor([a.nodes], device=dev))
    args = make_args(num_layers=a.layers, hidden_channels=a.hidden, dropout=0.0, conv_encode_edge=True,
                     use_edge_attr=True, use_column="w", global_edge="none", gcn_aggr="softmax", block="res+",
                     norm="layer", graph_pooling="mean", pathway_readout=None)
    torch.manual_seed(0)
    model = get_model("deepergcn")(args).to(dev)
    if a.dtype == "bf16":                      # bf16 weights + activations; aggregation arithmetic stays fp32
        model.to(torch.bfloat16)
        batch.x = batch.x.to(torch.bfloat16)
        batch.age = batch.age.to(torch.bfloat16)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(batch)
        (-torch.log(out[:, 0].float() + 1e-9)).sum().backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    if a.host_profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(12)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:      # autograd-thread ops too
            step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60),
              file=sys.stderr)
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    ops.KERNEL_TIMER = None
    res = {"config": "N=%d E=%d d=%d layers=%d %s" % (a.nodes, a.edges, a.hidden, a.layers, a.dtype),
           "step_ms": dt * 1e3, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "kernels": {}}
    for name, d in timer.summary().items():
        gbs = d["bytes"] / (d["avg_ms"] * 1e-3) / 1e9
        res["kernels"][name] = {"launches": d["launches"], "avg_ms": round(d["avg_ms"], 4),
                                "algorithmic_GB": round(d["bytes"] / 1e9, 3), "GBps": round(gbs, 1),
                                "frac_of_8TBps": round(gbs / 8000.0, 3)}

    # DiffPool at the stress size (pooled graph of 4096 nodes, 1024 clusters): bf16 -> the matrix-core product chain of
    # csrc/diffpool_large.hip (tools/bench_diffpool.py has the full report); fp32 -> the same chain with three-term products
    P, K, C = 4096, 1024, a.hidden
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    peak = 2500.0 if a.dtype == "bf16" else 157.3                 # TFLOP/s dense: bf16 MFMA / fp32 matrix (MI355X_MICROARCH.md)
    z = torch.randn(1, P, C, device=dev).to(dt).requires_grad_(True)
    s = torch.randn(1, P, K, device=dev).to(dt).requires_grad_(True)
    adj = torch.rand(P, P, device=dev).to(dt)

    def pool_fwd():
        return dense_diff_pool(z, adj, s)

    def pool():
        x, aa, l, e = pool_fwd()
        (x.float().sum() + aa.float().sum() + l.float() + e.float()).backward()

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    from mlgnn.dense import diff_pool_large_supported
    large_fp32 = a.dtype != "bf16" and bool(diff_pool_large_supported(z, adj.unsqueeze(0), s))
    with torch.no_grad():
        dtf = timed(pool_fwd)
    dtp = timed(pool)
    flop_fwd = 2 * K * P * C + 2 * K * P * P + 2 * K * K * P + 2 * P * P * K
    res["diffpool_4096_1024"] = {"dtype": a.dtype, "fwd_ms": dtf * 1e3, "fwd_bwd_ms": dtp * 1e3, "fwd_GFLOP": flop_fwd / 1e9,
                                 "fwd_TFLOPs": flop_fwd / dtf / 1e12, "fwd_MFMA_utilisation": flop_fwd / dtf / 1e12 / peak,
                                 "approx_TFLOPs_fwd_bwd": 3 * flop_fwd / dtp / 1e12,
                                 "dense_peak_TFLOPs": peak,
                                 "path": "csrc/diffpool_large.hip + gemm_nt.hip (bf16 MFMA)" if a.dtype == "bf16"
                                 else ("gemm_nt.hip, every product as three bf16 terms (mlgnn_diffpool_large_f32_fwd / _bwd)"
                                       if large_fp32 else "library GEMMs (fp32)"),
                                 "note": "FLOP counted in the reference's formulation (incl. S S^T); see tools/bench_diffpool.py"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
