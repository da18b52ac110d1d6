#!/usr/bin/env python3
"""Development tool: one configuration of tools/fuzz_aggregate.py with the oracle in fp64 next to the fp32 oracle the
sweep compares against -- tells a rounding-limited scalar gradient (d loss / dt sums N*d cancelling terms) from a wrong one.
`python tools/fuzz_case_fp64.py N E d aggr seed t`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    import torch
    import test_aggregate_gpu as T
    from mlgnn import CSRGraph, gen_aggregate
    N, E, d, aggr, seed, t = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), float(sys.argv[6])
    gen = torch.Generator().manual_seed(seed)
    ei = T._graph(gen, N, E, True)
    x = torch.randn(N, d, generator=gen)
    torch.rand(E, generator=gen); torch.randn(d, generator=gen); torch.randn(d, generator=gen); torch.randn(E, d, generator=gen)
    cot = torch.randn(N, d, generator=gen)
    res = {}
    for name, dt in (("fp32 oracle", torch.float32), ("fp64 oracle", torch.float64)):
        xx, tt = x.to(dt).requires_grad_(True), torch.tensor([t], dtype=dt, requires_grad=True)
        msg = torch.relu(xx[ei[0]]) + 1e-7
        ref = T.G.gen_aggregate(msg, ei[1], N, aggr, t=tt, learn_t=True, p=3.0)
        res[name] = float(torch.autograd.grad((ref * cot.to(dt)).sum(), tt)[0])
    dev = torch.device("cuda:0")
    xg, tg = x.to(dev).requires_grad_(True), torch.tensor([t], device=dev, requires_grad=True)
    out = gen_aggregate(xg, CSRGraph(ei.to(dev), N), None, aggr=aggr, t=tg, p=3.0, learn_t=True, learn_p=False)
    res["HIP"] = float(torch.autograd.grad((out * cot.to(dev)).sum(), tg)[0])
    for k, v in res.items():
        print("%-12s grad t = %.9f   (diff to fp64 oracle %.3e)" % (k, v, v - res["fp64 oracle"]))


if __name__ == "__main__":
    main()
