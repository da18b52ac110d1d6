#!/usr/bin/env python3
"""Time the pooled levels of the config-2 workload alone (projection pooling output -> DiffPool
146 -> 37 -> 10 -> head), forward + backward, with CUDA events.  Development tool."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn.workload import N_PATHWAYS, pathway_adjacency  # noqa: E402
from models.diff_pooling import DiffPool  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, hidden = 64 * 6, 128
    dp = DiffPool(hidden, None, N_PATHWAYS, 2, 32, 64,
                  SimpleNamespace(pooling_type="correlation", after_pooling_layer=1)).to(dev)
    adj = pathway_adjacency().to(dev)
    z = torch.randn(B, N_PATHWAYS, hidden, device=dev, requires_grad=True)

    def step():
        out, link, ent = dp(z, adj)
        (out.sum() + link + ent).backward()

    for _ in range(3):
        step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        step()
    e.record()
    torch.cuda.synchronize()
    print("DiffPool 146->37->10 on [%d,146,%d]: %.3f ms fwd+bwd" % (B, hidden, s.elapsed_time(e) / 10))


if __name__ == "__main__":
    main()
