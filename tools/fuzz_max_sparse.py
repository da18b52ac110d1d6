#!/usr/bin/env python3
"""Development tool: randomised sweep of the max aggregator's backward from compact winner lists (csrc/max_sparse.hip,
the destination-side table gradients of csrc/embedding.hip) against the general by-source backward on the same inputs:
random sizes, widths 32 .. 256, table sizes from none to more rows than edges, with and without the identity branch,
one to three layers.  `python tools/fuzz_max_sparse.py [cases] [seed]`; prints the first failing configuration."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def one(cfg):
    from _util import assert_close
    from mlgnn import CSRGraph, TableEdge, gen_aggregate, ops
    N, E, d, T, add_root, layers, seed = (cfg[k] for k in ("N", "E", "d", "T", "add_root", "layers", "seed"))
    gen = torch.Generator().manual_seed(seed)
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, max(N - 2, 1), (E,), generator=gen)
    k = min(8, E)
    src[:k] = dst[:k]
    ei = torch.stack([src, dst])
    x0 = torch.randn(N, d, generator=gen)
    table0 = torch.randn(max(T, 1), d, generator=gen) * 0.5
    idx = torch.randint(0, max(T, 1), (E,), generator=gen)
    cot = torch.randn(N, d, generator=gen)
    graph = CSRGraph(ei.to(DEV), N)
    graph.hub_tables("dst")
    torch.cuda.synchronize()
    short = graph.known_short_rows()

    def run(sparse):
        ops.SPARSE_MAX = sparse
        xd, td = x0.to(DEV).requires_grad_(True), table0.to(DEV).requires_grad_(True)
        te = TableEdge(td, idx.to(DEV)) if T else None
        h = xd
        for _ in range(layers):
            h = gen_aggregate(h, graph, te, aggr="max", add_root=add_root) * 0.5
        return torch.autograd.grad((h * cot.to(DEV)).sum(), [xd, td] if T else [xd])

    ref = run(False)
    before = ops.SPARSE_MAX_STATS["calls"]
    got = run(True)
    took = ops.SPARSE_MAX_STATS["calls"] - before
    assert took == (layers if short else 0), "sparse path taken %d times, rows short: %s" % (took, short)
    again = run(True)
    for a, b in zip(got, again):
        assert torch.equal(a, b), "not repeatable"
    assert_close(got[0], ref[0], 2e-6, "grad x", elementwise=True)
    if T:
        assert_close(got[1], ref[1], 2e-6, "grad table")
    return short


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    taken = 0
    for i in range(n_cases):
        N = rng.choice([3, 17, 64, 65, 130, 257, 600, 1500, 4001])
        E = rng.choice([1, 7, 33, 64, 65, 300, 1000, 4097, 9000, 40000])
        d = rng.choice([32, 36, 48, 64, 100, 128, 132, 200, 256])
        T = rng.choice([0, 0, 1, 5, 36, 37, 400, 5000, 60000])
        cfg = dict(N=N, E=E, d=d, T=T, add_root=rng.random() < 0.5, layers=rng.choice([1, 1, 2, 3]), seed=i)
        try:
            taken += bool(one(cfg))
        except Exception as exc:              # noqa: BLE001 -- report the configuration, then fail
            print("FAILED case %d: %r\n%s: %s" % (i, cfg, type(exc).__name__, exc))
            raise SystemExit(1)
        if (i + 1) % 25 == 0:
            print("%d cases ok (%d through the compact path)" % (i + 1, taken), flush=True)
    print("all %d cases ok (%d through the compact path)" % (n_cases, taken))


if __name__ == "__main__":
    main()
