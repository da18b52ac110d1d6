#!/usr/bin/env python3
"""Development tool: DeeperGCN training step with the reference's DEFAULT flags (opt.py: 3 layers, d=128, res+,
LayerNorm, gcn_aggr=max, global_edge='onehot' -> nn.Embedding edge types -> materialised [E,d] edge embedding shared
by all layers) on synthetic ER graphs of BASELINE configs[1] size.  Prints ms per step and graphs/s."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import make_args  # noqa: E402
from models import get_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=160000)
    ap.add_argument("--edge-types", type=int, default=20000)
    ap.add_argument("--aggr", default="max")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, n, e = a.graphs, a.nodes, a.edges
    gen = torch.Generator(device=dev).manual_seed(0)
    offs = torch.arange(B, device=dev)[:, None] * n
    ei = torch.stack([(torch.randint(0, n, (B, e), generator=gen, device=dev) + offs).reshape(-1),
                      (torch.randint(0, n, (B, e), generator=gen, device=dev) + offs).reshape(-1)])
    batch = SimpleNamespace(x=torch.randn(B * n, 3, device=dev), edge_index=ei,
                            edge_attr=torch.randint(0, a.edge_types, (B * e, 1), generator=gen, device=dev).float(),
                            batch=torch.arange(B, device=dev).repeat_interleave(n), age=torch.rand(B, device=dev),
                            pathway_node_attr=None, node_size=torch.full((B,), n, device=dev))
    args = make_args(use_edge_attr=True, global_edge="onehot", pathway_edge_num=a.edge_types, gcn_aggr=a.aggr,
                     dropout=0.5, pathway_readout=None)
    torch.manual_seed(0)
    model = get_model("deepergcn")(args).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    y = torch.nn.functional.one_hot(torch.randint(0, 2, (B,), device=dev), 2).float()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.binary_cross_entropy(model(batch), y)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("DeeperGCN default flags (aggr=%s, onehot edge types=%d): %.2f ms/step, %.0f graphs/s (B=%d)"
          % (a.aggr, a.edge_types, dt * 1e3, B / dt, B))


if __name__ == "__main__":
    main()
