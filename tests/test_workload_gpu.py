"""Composite 3-level GNN (bench workload) on the HIP path vs the CPU oracle executing the
reference's literal op sequence, forward + every parameter gradient.  1e-4 fp32."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("aggr,kw", [("softmax", {}), ("max", {}), ("mean", {}),
                                     ("softmax", dict(learn_t=True, t=0.7, msg_norm=True))])
def test_three_level_gnn_matches_oracle(aggr, kw):
    from mlgnn import workload as W
    from oracle import workload as OW
    torch.manual_seed(3)
    n, e, members, hidden = 300, 2400, 900, 32
    model = W.ThreeLevelGNN(hidden=hidden, aggr=aggr, n_members=members, **kw)
    match, seg = W.membership(n, members)
    cpu_batch = W.collate([0, 1, 2], n, e, match, seg, "cpu")
    sd = {k: v.detach().clone().requires_grad_(k != "pathway_adj") for k, v in model.state_dict().items()}
    ref_loss = OW.training_loss(sd, cpu_batch, aggr=aggr, **kw)
    names = [k for k, v in sd.items() if v.requires_grad]
    ref_grads = dict(zip(names, torch.autograd.grad(ref_loss, [sd[k] for k in names], allow_unused=True)))
    ref_pred = OW.three_level_forward(sd, cpu_batch, aggr=aggr, **kw)[0]

    model.to("cuda:0")
    batch = W.collate([0, 1, 2], n, e, match, seg, "cuda:0")
    pred, _, _ = model(batch)
    assert_close(pred, ref_pred, 1e-4, "pred")
    loss = W.training_loss(model, batch)
    assert_close(loss, ref_loss, 1e-4, "loss")
    loss.backward()
    for name, p in model.named_parameters():
        g = ref_grads[name]
        g = torch.zeros_like(sd[name]) if g is None else g
        got = p.grad if p.grad is not None else torch.zeros_like(p)     # DiffPool.initial_embed is never called
        assert_close(got, g, 1e-4, "grad " + name)


def test_full_size_properties():
    """BASELINE config-2 size (one graph batch of 8): size-independent properties of the
    aggregation -- permutation invariance of the edge list, linearity of 'add', mean = add/deg,
    softmax output bounded by [min, max] of the messages."""
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
    from mlgnn import workload as W
    dev = "cuda:0"
    match, seg = W.membership(10000, 100)
    b = W.collate(list(range(8)), 10000, 160000, match, seg, dev)
    N = b.x.shape[0]
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(N, 128, device=dev, generator=gen)
    u = torch.randn(128, device=dev, generator=gen) * 0.3
    v = torch.randn(128, device=dev, generator=gen) * 0.1
    g = CSRGraph(b.edge_index, N)
    perm = torch.randperm(b.edge_index.shape[1], device=dev, generator=gen)
    g2 = CSRGraph(b.edge_index[:, perm], N)
    e1 = RankOneEdge(b.edge_attr[:, 0], u, v)
    e2 = RankOneEdge(b.edge_attr[perm, 0], u, v)
    add = gen_aggregate(x, g, e1, aggr="add")
    assert_close(gen_aggregate(x, g2, e2, aggr="add"), add, 1e-4, "edge-order invariance (add)")
    assert_close(gen_aggregate(x, g2, e2, aggr="max"), gen_aggregate(x, g, e1, aggr="max"), 0.0, "edge-order invariance (max)")
    sm = gen_aggregate(x, g, e1, aggr="softmax")
    assert_close(gen_aggregate(x, g2, e2, aggr="softmax"), sm, 1e-4, "edge-order invariance (softmax)")
    mean = gen_aggregate(x, g, e1, aggr="mean")
    assert_close(mean * g.in_degree.clamp(min=1)[:, None], add, 1e-4, "mean * deg = add")
    mx = gen_aggregate(x, g, e1, aggr="max")
    has = (g.in_degree > 0)[:, None]
    assert bool(((sm <= mx + 1e-5) | ~has).all()) and bool(((sm >= 1e-7 - 1e-9) | ~has).all())
    # block-diagonal batch: graph 3 alone gives the same rows as inside the batch
    one = W.collate([3], 10000, 160000, match, seg, dev)
    g1 = CSRGraph(one.edge_index, 10000)
    xs = x[30000:40000].contiguous()
    alone = gen_aggregate(xs, g1, RankOneEdge(one.edge_attr[:, 0], u, v), aggr="softmax")
    assert_close(alone, sm[30000:40000], 1e-5, "graphs in a batch are independent")


def test_three_level_gnn_full_size_graphs_match_oracle():
    """BASELINE configs[1] at its real per-graph size (N=10 000, E=160 000, d=128, G=25 000), two graphs:
    the oracle's literal [E,d] op sequence finishes in a few seconds on the CPU."""
    from mlgnn import workload as W
    from oracle import workload as OW
    torch.manual_seed(11)
    n, e, members = 10000, 160000, 25000
    model = W.ThreeLevelGNN(hidden=128, aggr="softmax", n_members=members)
    match, seg = W.membership(n, members)
    cpu_batch = W.collate([5, 6], n, e, match, seg, "cpu")
    sd = {k: v.detach().clone().requires_grad_(k != "pathway_adj") for k, v in model.state_dict().items()}
    ref_loss = OW.training_loss(sd, cpu_batch, aggr="softmax")
    names = [k for k, v in sd.items() if v.requires_grad]
    ref_grads = dict(zip(names, torch.autograd.grad(ref_loss, [sd[k] for k in names], allow_unused=True)))
    model.to("cuda:0")
    loss = W.training_loss(model, W.collate([5, 6], n, e, match, seg, "cuda:0"))
    assert_close(loss, ref_loss, 1e-4, "loss")
    loss.backward()
    for name, p in model.named_parameters():
        g = ref_grads[name]
        g = torch.zeros_like(sd[name]) if g is None else g
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(got, g, 1e-4, "grad " + name)


def test_row_maxima_travel_with_the_activations():
    """The aggregation / LayerNorm kernels hand max |row| to the Linear that consumes their output, forward and
    backward, so the tall GEMM does not stream its operand twice: in one training step of the 3-layer model every
    MLP GEMM except the node encoder's finds its row maxima attached."""
    from mlgnn import dense
    from mlgnn.workload import ThreeLevelGNN, collate, membership, training_loss
    dev = "cuda:0"
    torch.manual_seed(0)
    match, seg = membership(2500, 3000)
    batch = collate(range(4), 2500, 30000, match, seg, device=dev)
    model = ThreeLevelGNN(hidden=64, num_layers=3, n_members=3000).to(dev)
    dense.ROW_MAX_STATS["given"] = dense.ROW_MAX_STATS["computed"] = 0
    training_loss(model, batch).backward()
    st = dict(dense.ROW_MAX_STATS)
    # 3 layers x (2 forward + 2 input-gradient GEMMs) = 12 tall GEMMs; the first layer's aggregation output and
    # every LayerNorm output / gradient carry their maxima
    assert st["given"] >= 10, st
