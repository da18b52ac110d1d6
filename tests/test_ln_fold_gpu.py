"""The res+ block's pre-conv ``y = relu(LayerNorm(h))`` (reference: models/deepergcn.py:236-241) taken backward inside the
row epilogue of the aggregation that consumes ``y`` (``mlgnn_csr_aggregate_bwd_ln``, csrc/aggregate_bwd.hip LNB) against
the two-pass form: the plain aggregation backward followed by torch's own LayerNorm / ReLU backward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _graph(n, e, seed, dev):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(dev)
    return ei


@pytest.mark.parametrize("aggr", ["softmax", "max", "mean"])
@pytest.mark.parametrize("d,relu,with_extra", [(128, True, True), (64, True, False), (256, False, True), (128, False, False)])
def test_epilogue_equals_two_passes(aggr, d, relu, with_extra):
    from mlgnn import CSRGraph, RankOneEdge, gen_aggregate, ops
    dev = torch.device("cuda:0")
    n, e = 3000, 40000
    graph = CSRGraph(_graph(n, e, 5, dev), n)
    graph.hub_tables("src")
    torch.cuda.synchronize()                    # the graph is known hub-free on the host from here on
    g = torch.Generator().manual_seed(9)
    h = (torch.randn(n, d, generator=g) * 1.5 + 0.3).to(dev)
    gamma = (torch.rand(d, generator=g) + 0.5).to(dev)
    beta = (torch.randn(d, generator=g) * 0.2).to(dev)
    go = torch.randn(n, d, generator=g).to(dev)
    extra = torch.randn(n, d, generator=g).to(dev) if with_extra else None
    edge = RankOneEdge(torch.rand(e, generator=g).to(dev), (torch.randn(d, 1, generator=g) * 0.3).to(dev),
                       (torch.randn(d, generator=g) * 0.1).to(dev))
    eps = 1e-5
    mean = h.mean(1)
    rstd = torch.rsqrt(h.var(1, unbiased=False) + eps)

    def run(fold):
        old = ops.LN_FOLD
        ops.LN_FOLD = fold
        try:
            hh = h.clone().requires_grad_(True)
            gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y = F.layer_norm(hh, (d,), gm, bt, eps)
            y = torch.relu(y) if relu else y
            yt = y.detach().clone().requires_grad_(True)
            tag = ops.PostLN(h, mean, rstd, gamma, beta, relu)
            tag.extra = extra
            yt._mlgnn_post_ln = tag
            out = gen_aggregate(yt, graph, edge, aggr=aggr, add_root=True)
            out.backward(go)
            return hh, gm, bt, y, yt, tag
        finally:
            ops.LN_FOLD = old

    hh, gm, bt, y, yt, tag = run(False)
    assert tag.folded is None and yt.grad is not None
    y.backward(yt.grad)
    ref_h = hh.grad + (extra if extra is not None else 0)
    ref_g, ref_b = gm.grad, bt.grad

    before = ops.LN_FOLD_STATS["folded"]
    _, _, _, _, yt2, tag2 = run(True)
    assert ops.LN_FOLD_STATS["folded"] == before + 1
    assert yt2.grad is None and tag2.folded is not None and tag2.extra_used == with_extra
    gx, dg, db, rmax = tag2.folded
    scale = float(ref_h.abs().max())
    assert float((gx - ref_h).abs().max()) <= 2e-5 * scale
    assert float((dg - ref_g).abs().max()) <= 1e-4 * float(ref_g.abs().max())
    assert float((db - ref_b).abs().max()) <= 1e-4 * float(ref_b.abs().max())
    assert torch.equal(rmax, gx.abs().amax(1))


def test_res_plus_stack_gradients_identical_with_and_without_fold():
    """Three GENConv layers in the res+ arrangement at a size where the fused MLP (and so the fold) is active: every
    parameter gradient with the fold equals the gradient without it."""
    from mlgnn import ops
    from mlgnn import workload as W
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    n, e, members = 6000, 60000, 9000
    model = W.ThreeLevelGNN(hidden=128, aggr="softmax", n_members=members).to(dev)
    match, seg = W.membership(n, members)
    batch = W.collate([0, 1], n, e, match, seg, dev)
    from mlgnn import CSRGraph
    batch.csr = CSRGraph(batch.edge_index, batch.x.shape[0])
    batch.csr.hub_tables("src")
    torch.cuda.synchronize()                    # known hub-free on the host: the fold needs rows finished by one wave

    def grads(fold):
        old = ops.LN_FOLD
        ops.LN_FOLD = fold
        try:
            model.zero_grad(set_to_none=True)
            f0 = ops.LN_FOLD_STATS["folded"]
            loss = W.training_loss(model, batch)
            loss.backward()
            return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, \
                ops.LN_FOLD_STATS["folded"] - f0
        finally:
            ops.LN_FOLD = old

    l0, g0, n0 = grads(False)
    l1, g1, n1 = grads(True)
    assert n0 == 0 and n1 == 2                  # layers 1 and 2 read a post-LayerNorm y; layer 0 reads the encoder output
    assert l0 == l1
    assert g0.keys() == g1.keys()
    for k in g0:
        ref = float(g0[k].abs().max())
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * max(ref, 1e-6), k
