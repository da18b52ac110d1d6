#!/usr/bin/env python3
"""Micro-benchmark of the CSR aggregation kernels at BASELINE config-2 size (per-GPU batch of
synthetic ER graphs, N=10k, E=160k per graph, d=128 fp32).  HIP-event timing on the launch stream;
prints algorithmic GB/s (SURVEY.md section 8d byte formula) per kernel.  Development tool."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn import CSRGraph, LowRankEdge, RankOneEdge, gen_aggregate  # noqa: E402


def er_batch(n_graphs, n, e, dev, seed=1000):
    gen = torch.Generator(device=dev).manual_seed(seed)
    src = torch.randint(0, n, (n_graphs, e), generator=gen, device=dev)
    dst = torch.randint(0, n, (n_graphs, e), generator=gen, device=dev)
    offs = torch.arange(n_graphs, device=dev)[:, None] * n
    return torch.stack([(src + offs).reshape(-1), (dst + offs).reshape(-1)])


def timed(fn, iters, warmup=3):
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
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=160000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, E, d = a.graphs * a.nodes, a.graphs * a.edges, a.d
    ei = er_batch(a.graphs, a.nodes, a.edges, dev)
    t_csr = timed(lambda: CSRGraph(ei, N), 3, 1)
    graph = CSRGraph(ei, N)
    x = torch.randn(N, d, device=dev)
    ea = torch.rand(E, device=dev)
    u = torch.randn(d, device=dev) * 0.5
    v = torch.randn(d, device=dev) * 0.1
    ef = torch.randn(E, d, device=dev) * 0.5
    ea7 = torch.rand(E, 7, device=dev)                       # the reference's default 7-column attribute
    u7 = torch.randn(d, 7, device=dev) * 0.3
    fwd_bytes = E * d * 4 + E * 4 + (N + 1) * 4 + E * 4 + N * d * 4
    print("N=%d E=%d d=%d  CSR build %.2f ms  fwd algorithmic bytes %.1f MB" % (N, E, d, t_csr, fwd_bytes / 1e6))
    for aggr in ("softmax", "max", "mean"):
        for kind in ("rank1", "rank7", "full"):
            xr = x.clone().requires_grad_(True)
            if kind == "rank1":
                ur, vr = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
                edge = RankOneEdge(ea, ur, vr)
                fb = fwd_bytes
            elif kind == "rank7":
                ur, vr = u7.clone().requires_grad_(True), v.clone().requires_grad_(True)
                edge = LowRankEdge(ea7, ur, vr)
                fb = fwd_bytes + E * 4 * 7                    # 8 padded scalars per edge instead of 1
            else:
                edge = ef.clone().requires_grad_(True)
                fb = fwd_bytes - E * 4 + E * d * 4 + E * 4
            with torch.no_grad():
                tf = timed(lambda: gen_aggregate(xr, graph, edge, aggr=aggr), a.iters)
            out = gen_aggregate(xr, graph, edge, aggr=aggr)
            go = torch.randn_like(out)
            tb = timed(lambda: torch.autograd.grad(out, [xr], go, retain_graph=True), a.iters)
            print("%-8s %-6s fwd %7.3f ms  %7.1f GB/s (%.1f%% of 8 TB/s) | bwd %7.3f ms" %
                  (aggr, kind, tf, fb / tf / 1e6, fb / tf / 1e6 / 80.0, tb))


if __name__ == "__main__":
    main()
