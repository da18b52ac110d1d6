#!/usr/bin/env python3
"""Development tool: which host-side ops of one training step of the bench workload launch device copies / fills
(``__amd_rocclr_copyBuffer`` / ``fillBufferAligned`` in the rocprofv3 summary).  Prints aten::copy_ / zero_ / fill_ /
clone / contiguous call sites (innermost frame inside this repo) with counts and device time."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))


def main():
    from mlgnn import workload as W
    from mlgnn.optim import FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, e, members, B = 10000, 160000, 25000, 64
    model = W.ThreeLevelGNN(hidden=128, num_layers=3, aggr="softmax", n_members=members).to(dev)
    opt = FlatAdam(model, lr=1e-3)
    match, seg = W.membership(n, members)
    batch = W.collate(list(range(B)), n, e, match, seg, dev)

    def step():
        batch.csr = None
        opt.zero_grad()
        loss = W.training_loss(model, batch)
        loss.backward()
        opt.bucket.collect()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::zero_", "aten::fill_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::_foreach_copy_",
            "aten::cat", "aten::index", "aten::to", "aten::_to_copy")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name not in want:
            continue
        site = "?"
        for fr in ev.stack:
            if ROOT in fr and "tools/find_copies" not in fr:
                site = fr.replace(ROOT + "/", "")
                break
        dt = getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0)
        k = (ev.name, site)
        agg[k][0] += 1
        agg[k][1] += dt
    for (name, site), (cnt, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-22s x%-3d %8.1f us  %s" % (name, cnt, dt, site))
    print("---- device kernels (top) ----")
    print(prof.key_averages().table(sort_by="self_device_time_total" if hasattr(torch.autograd.profiler_util.FunctionEventAvg, "self_device_time_total") else "self_cuda_time_total", row_limit=40, max_name_column_width=70))


if __name__ == "__main__":
    main()
