#!/usr/bin/env python3
"""Development tool: aggregation kernel time on a hub-heavy (GRN-like) graph vs an ER graph of the
same size (TCGA shape: 15 405 nodes and 60 000 edges per graph, B graphs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn import CSRGraph, RankOneEdge, gen_aggregate, weighted_mean_aggregate  # noqa: E402


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


def graph(B, n, e, hubs, hub_deg, dev, hub_is_source=True):
    gen = torch.Generator().manual_seed(1)
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    if hubs:
        idx = torch.arange(hubs * hub_deg)
        (src if hub_is_source else dst)[idx] = idx // hub_deg          # first `hubs` nodes are transcription factors
    ei = torch.cat([torch.stack([src, dst]) + b * n for b in range(B)], dim=1)
    return ei.to(dev)


def main():
    import json
    from mlgnn import graph as graph_mod
    dev = torch.device("cuda:0")
    B, n, e, d = 32, 15405, 60000, 64
    rows = []
    print("HUB_CAP =", graph_mod.HUB_CAP)
    for name, hubs, deg, as_src in [("ER", 0, 0, True), ("20 hubs x 1500 out-edges", 20, 1500, True),
                                    ("20 hubs x 1500 in-edges", 20, 1500, False), ("2 hubs x 15000 out", 2, 15000, True)]:
        ei = graph(B, n, e, hubs, deg, dev, as_src)
        N = B * n
        g = CSRGraph(ei, N)
        x = torch.randn(N, d, device=dev, requires_grad=True)
        w = torch.rand(ei.shape[1], device=dev)
        tf = timed(lambda: weighted_mean_aggregate(x.detach(), g, w))
        out = weighted_mean_aggregate(x, g, w)
        go = torch.randn_like(out)
        tb = timed(lambda: torch.autograd.grad(out, [x], go, retain_graph=True))
        u, v = torch.randn(d, device=dev), torch.randn(d, device=dev)
        tsf = timed(lambda: gen_aggregate(x.detach(), g, RankOneEdge(w, u, v), aggr="softmax"))
        o2 = gen_aggregate(x, g, RankOneEdge(w, u, v), aggr="softmax")
        tsb = timed(lambda: torch.autograd.grad(o2, [x], go, retain_graph=True))
        print("%-28s SAGE mean fwd %.3f ms bwd %.3f ms | GEN softmax fwd %.3f ms bwd %.3f ms" % (name, tf, tb, tsf, tsb))
        rows.append({"graph": name, "sage_mean_fwd_ms": tf, "sage_mean_bwd_ms": tb, "gen_softmax_fwd_ms": tsf,
                     "gen_softmax_bwd_ms": tsb})
    er = rows[0]
    for r in rows:
        r["vs_ER"] = {k: round(r[k] / er[k], 2) for k in er if k.endswith("_ms")}
    print(json.dumps({"hub_cap": graph_mod.HUB_CAP, "config": "%d graphs x %d nodes x %d edges, d=%d, fp32" % (B, n, e, d),
                      "rows": rows}))


if __name__ == "__main__":
    main()
