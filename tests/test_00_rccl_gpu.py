"""Two ranks over RCCL (backend ``nccl``), one process per GPU: the data-parallel step of bench.py / train_harness.py
(flat gradient bucket -> one all-reduce -> fused Adam) gives every rank the parameters a single process gets from the
global batch.  Needs two GPUs: skipped on the one-GPU test boxes (the CPU twin is tests/test_dist_cpu.py, gloo).

The ranks are fresh interpreter processes (multiprocessing ``spawn``) started BEFORE this process touches the GPU --
this file sorts first for that reason, counting devices does not initialise HIP, and the test skips itself if something
already has."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(16, 16, generator=g), torch.randn(16, 4, generator=g)


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "multilevel-gnn_amd"))
    import torch.distributed as dist
    from mlgnn.dist import broadcast_parameters
    from mlgnn.optim import FlatAdam
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(50 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4)).to(dev)
    if rank == 0:
        model.load_state_dict(_net().state_dict())
    broadcast_parameters(model)
    opt = FlatAdam(model, lr=1e-2, weight_decay=1e-3, clip_grad_norm=5.0)
    x, y = _data()
    n = x.shape[0] // world
    xs, ys = x[rank * n:(rank + 1) * n].to(dev), y[rank * n:(rank + 1) * n].to(dev)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(xs), ys).backward()
        opt.bucket.collect()
        opt.bucket.all_reduce_mean()
        opt.step()
    torch.cuda.synchronize()
    q.put((rank, opt.flat_p.cpu()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_match_one_process():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds a GPU context; the ranks must be started before that")
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=300) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert [p.exitcode for p in procs] == [0] * world
    # single process, global batch (mean of equal shard means = global mean), torch's own optimizer
    ref = _net()
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-3)
    x, y = _data()
    for _ in range(3):
        topt.zero_grad()
        torch.nn.functional.mse_loss(ref(x), y).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        topt.step()
    want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    assert torch.equal(got[0], got[1])                                   # ranks stay in lock step
    assert float((got[0] - want).abs().max()) <= 1e-5
