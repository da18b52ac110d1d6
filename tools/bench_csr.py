#!/usr/bin/env python3
"""Topology build on its own (no step running next to it): COO -> both CSR orderings + the rank-1 edge table of a
BASELINE configs[1] batch (64 ER graphs, 10k nodes / 160k edges each).  HIP-event time per build; run it under
``rocprofv3 --kernel-trace --stats`` for the per-kernel view (tools/kernel_stats.py).  Development tool."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn import CSRGraph  # noqa: E402


def er_batch(n_graphs, n, e, dev, seed=1000):
    gen = torch.Generator(device=dev).manual_seed(seed)
    src = torch.randint(0, n, (n_graphs, e), generator=gen, device=dev)
    dst = torch.randint(0, n, (n_graphs, e), generator=gen, device=dev)
    offs = torch.arange(n_graphs, device=dev)[:, None] * n
    return torch.stack([(src + offs).reshape(-1), (dst + offs).reshape(-1)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=160000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shuffle", action="store_true", help="edge list in random order (no block-diagonal locality)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, E = a.graphs * a.nodes, a.graphs * a.edges
    ei = er_batch(a.graphs, a.nodes, a.edges, dev)
    if a.shuffle:
        ei = ei[:, torch.randperm(E, device=dev)].contiguous()
    attr = torch.rand(E, 1, device=dev)

    def build():
        g = CSRGraph(ei, N)
        g.edge_table(attr, 1)
        return g

    for _ in range(3):
        build()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(a.iters):
        build()
    e.record()
    torch.cuda.synchronize()
    print(json.dumps({"workload": "%d graphs x %d nodes / %d edges%s" % (a.graphs, a.nodes, a.edges,
                                                                          ", shuffled" if a.shuffle else ""),
                      "csr_build_plus_edge_table_ms": s.elapsed_time(e) / a.iters}))


if __name__ == "__main__":
    main()
