"""Backward of the max aggregator from compact winner lists (csrc/max_sparse.hip: mlgnn_max_winners, mlgnn_max_sparse_bwd,
mlgnn_max_sparse_table_grad) -- the reference's DEFAULT aggregator (opt.py:144; torch_message.py:46-47) with and without
the edge-type table of global_edge='onehot' (deepergcn.py:103-104) -- against the general by-source backward on the same
inputs and against the fp64 oracle of the reference's formulation."""
import pytest
import torch

from _util import assert_close
from oracle import gcn_lib as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _short_graph(gen, N, E, max_deg=None):
    """ER-like edge list with self loops, duplicates and two isolated nodes; no row beyond 256 edges in either direction."""
    src = torch.randint(0, N, (E,), generator=gen)
    dst = torch.randint(0, N - 2, (E,), generator=gen)
    src[:8] = dst[:8]
    src[8:16], dst[8:16] = src[16:24], dst[16:24]
    if max_deg:                                            # one destination with exactly max_deg incoming edges, one such source
        dst[dst == 5] = 6
        dst[100:100 + max_deg] = 5
        src[src == 9] = 10
        src[400:400 + max_deg] = 9
    return torch.stack([src, dst])


def _ready(graph):
    graph.hub_tables("dst")
    torch.cuda.synchronize()
    assert graph.known_short_rows()


@pytest.mark.parametrize("d,T,add_root,layers", [(128, 0, True, 1), (128, 0, False, 2), (128, 8, True, 3), (128, 5000, True, 3),
                                                 (64, 3000, False, 2), (100, 700, True, 1), (256, 40000, True, 2), (256, 40000, True, 1),
                                                 (32, 0, True, 2), (36, 90, False, 1), (128, 5000, False, 1)])
def test_sparse_max_backward_matches_general_and_oracle(d, T, add_root, layers, monkeypatch):
    from mlgnn import CSRGraph, TableEdge, gen_aggregate, ops
    gen = torch.Generator().manual_seed(300 + d + T)
    N, E = 2999, 30000
    ei = _short_graph(gen, N, E, max_deg=256 if d == 128 else 40)
    # inputs on a 2^-12 grid: x_j + e is then exact in fp32 and in fp64, both pick the same winners (first maximal edge on
    # exact ties), and the comparison with the fp64 oracle cannot trip over a near-tie resolved differently
    x0 = torch.round(torch.randn(N, d, generator=gen) * 4096) / 4096
    table0 = torch.round(torch.randn(max(T, 1), d, generator=gen) * 2048) / 4096
    idx = torch.randint(0, max(T - 1, 1), (E,), generator=gen)
    cot = torch.randn(N, d, generator=gen)
    graph = CSRGraph(ei.to(DEV), N)
    _ready(graph)

    def run(sparse):
        monkeypatch.setattr(ops, "SPARSE_MAX", sparse)
        xd, td = x0.to(DEV).requires_grad_(True), table0.to(DEV).requires_grad_(True)
        te = TableEdge(td, idx.to(DEV)) if T else None
        h = xd
        for _ in range(layers):
            h = gen_aggregate(h, graph, te, aggr="max", add_root=add_root) * 0.5
        return torch.autograd.grad((h * cot.to(DEV)).sum(), [xd, td] if T else [xd])

    ref = run(False)
    before = dict(ops.SPARSE_MAX_STATS)
    got = run(True)
    assert ops.SPARSE_MAX_STATS["calls"] == before["calls"] + layers
    if T > 36:
        assert ops.SPARSE_MAX_STATS["table"] == before["table"] + layers
    again = run(True)
    for a, b in zip(got, again):
        assert torch.equal(a, b)                           # no atomics between workgroups: bitwise repeatable
    assert_close(got[0], ref[0], 2e-6, "grad x vs the general backward", elementwise=True)
    if T:
        assert_close(got[1], ref[1], 2e-6, "table gradient vs the general backward")
        if T > 1:
            assert not bool(got[1][T - 1].any())           # a row no edge reads
    if layers > 1:
        return                                             # (deeper inputs are no longer on the grid; the general backward,
                                                           # checked against the oracle in test_aggregate_gpu.py, is the reference)
    # the reference's formulation in fp64
    xr, tr = x0.double().requires_grad_(True), table0.double().requires_grad_(True)
    h = xr
    for _ in range(layers):
        e = tr[idx] if T else 0
        m = G.gen_aggregate(torch.relu(h[ei[0]] + e) + 1e-7, ei[1], N, "max")
        h = ((h + m) if add_root else m) * 0.5
    want = torch.autograd.grad((h * cot.double()).sum(), [xr, tr] if T else [xr])
    assert_close(got[0], want[0], 1e-5, "grad x vs oracle", elementwise=True)
    if T:
        assert_close(got[1], want[1], 1e-5, "grad table vs oracle")


def test_sparse_max_backward_is_not_taken_with_long_rows(monkeypatch):
    from mlgnn import CSRGraph, gen_aggregate, ops
    gen = torch.Generator().manual_seed(5)
    N, E, d = 2000, 20000, 64
    ei = _short_graph(gen, N, E)
    ei[1, 1000:1300] = 3                                   # 300 incoming edges: beyond the cap
    graph = CSRGraph(ei.to(DEV), N)
    graph.hub_tables("dst")
    torch.cuda.synchronize()
    assert not graph.known_short_rows()
    x = torch.randn(N, d, generator=gen).to(DEV).requires_grad_(True)
    before = ops.SPARSE_MAX_STATS["calls"]
    gen_aggregate(x, graph, None, aggr="max").sum().backward()
    assert ops.SPARSE_MAX_STATS["calls"] == before and x.grad is not None


def test_winner_runs_partition_the_channels():
    """mlgnn_max_winners through the C ABI: the runs of a row's edges tile its records (each run on an even record), every
    (value, channel) pair is a winner of that edge with the cotangent of its channel, and channels without a winner appear
    in no run."""
    from mlgnn import CSRGraph, _lib
    gen = torch.Generator().manual_seed(11)
    N, E, d = 500, 6000, 128
    ei = _short_graph(gen, N, E)
    graph = CSRGraph(ei.to(DEV), N)
    go = torch.randn(N, d, generator=gen).to(DEV)
    rowptr = graph.rowptr.cpu()
    deg = (rowptr[1:] - rowptr[:-1])
    argmax = torch.full((N, d), -1, dtype=torch.int32)
    for i in range(N):
        if deg[i] > 0:
            a = torch.randint(0, int(deg[i]), (d,), generator=gen) + int(rowptr[i])
            a[torch.rand(d, generator=gen) < 0.1] = -1
            argmax[i] = a.int()
    n_rec = int(_lib.lib.mlgnn_max_sparse_records(N, d, E))
    recs = torch.zeros((n_rec, 2), dtype=torch.int32, device=DEV)
    meta = torch.zeros((E, 2), dtype=torch.int32, device=DEV)
    rc = _lib.lib.mlgnn_max_winners(go.data_ptr(), argmax.to(DEV).data_ptr(), graph.rowptr.data_ptr(), recs.data_ptr(),
                                    meta.data_ptr(), N, d, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    recs, meta, go = recs.cpu(), meta.cpu().long(), go.cpu()
    wval, wch = recs[:, 0].contiguous().view(torch.float32), recs[:, 1].long()
    for i in range(0, N, 7):
        at = (i * (d + 2) + int(rowptr[i]) + 1) // 2 * 2                 # the row's first record (even)
        seen, total = torch.zeros(d, dtype=torch.bool), 0
        for p in range(int(rowptr[i]), int(rowptr[i + 1])):
            off, cnt = int(meta[p, 0]), int(meta[p, 1])
            assert off == at and off % 2 == 0
            ch = wch[off:off + cnt]
            assert bool((argmax[i, ch] == p).all()) and not bool(seen[ch].any())
            assert torch.equal(wval[off:off + cnt], go[i, ch])
            assert cnt == int((argmax[i] == p).sum())
            seen[ch] = True
            at += (cnt + 1) // 2 * 2
            total += cnt
        assert total == int((argmax[i] >= 0).sum())
