"""The fused GraphSAGE layer (mlgnn/sage.py: aggregation + ONE dual-operand GEMM with the LeakyReLU / value-mask
epilogue; csrc/tallgemm.hip DUAL, csrc/sage.hip) against the CPU oracle's restatement of the reference's per-edge form
(models/gcn_lib/sparse/torch_vertex.py:269-304) and against the unfused operator sequence, at row counts past the
8192-row threshold of the tall kernels (the golden SAGE fixtures are smaller and take the unfused path)."""
import pytest
import torch

from _util import assert_close
from oracle import gcn_lib as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, e, gen):
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    dst[:50] = src[:50]                                   # self loops that the conv must drop
    dst[50:400] = 3                                       # a hub row (350 in-edges: past the 64-edge cap of the short-row kernels)
    src[400:700] = 7                                      # ... and a node with 300 out-edges (the backward's long row)
    dst[700:770] = 11                                     # 70 in-edges: just past the cap
    w = torch.rand(e, 1, generator=gen) * 2 - 1           # weights in [-1, 1] incl. negative ones (multiloader.py:671)
    return torch.stack([src, dst]), w


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 32), (64, 64), (128, 32), (32, 32)])
@pytest.mark.parametrize("kind,act", [("sage", "leakyrelu"), ("rsage", "leakyrelu"), ("sage", "relu")])
def test_fused_layer_matches_oracle(cin, cout, kind, act):
    from mlgnn import sage as S
    from models.gcn_lib.sparse.torch_vertex import GraphConv
    gen = torch.Generator().manual_seed(cin * 131 + cout + len(kind))
    n, e = 9001, 40000
    torch.manual_seed(5)
    conv = GraphConv(cin, cout, conv=kind, act=act, mlp_norm="none")
    ei, w = _graph(n, e, gen)
    x = torch.randn(n, cin, generator=gen)
    mask = torch.randn(n, generator=gen)
    mask[::7] = 0.0                                       # zero and negative mask values (the mask is the raw input value)
    cot = torch.randn(n, cout, generator=gen)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in conv.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    ref = G.sageconv(xr, ei, w, sd, "gconv.", act_name=act, relative=(kind == "rsage")) * mask[:, None]
    names = [k for k in sd if "lin_l" not in k]
    g_ref = dict(zip(["x"] + names, torch.autograd.grad((ref * cot).sum(), [xr] + [sd[k] for k in names])))

    conv.to(DEV)
    before = S.STATS["fused"]
    xg = x.to(DEV).requires_grad_(True)
    out = conv(xg, ei.to(DEV), w.to(DEV), row_scale=mask.to(DEV))
    assert S.STATS["fused"] == before + 1                 # the fused layer ran
    assert_close(out, ref, 1e-4, "out", elementwise=True)
    (out * cot.to(DEV)).sum().backward()
    assert_close(xg.grad, g_ref["x"], 1e-4, "grad x", elementwise=True)
    for k, p in conv.named_parameters():
        if "lin_l" in k:
            assert p.grad is None                         # dead in the reference too
            continue
        assert_close(p.grad, g_ref[k], 1e-4, "grad " + k)
    # the row maxima that travel with the result
    from mlgnn.ops import row_max_of
    assert torch.allclose(row_max_of(out), out.detach().abs().amax(1))


def test_fused_layer_equals_unfused_sequence(monkeypatch):
    """Same layer through the separate operators (aggregate, lin_r GEMM, cat, Linear, LeakyReLU, mask): outputs and all
    gradients agree to fp32 noise; two fused runs are bitwise equal (no atomics anywhere)."""
    from models.gcn_lib.sparse import torch_vertex as TV
    gen = torch.Generator().manual_seed(77)
    n, e, cin, cout = 20000, 90000, 64, 32
    torch.manual_seed(6)
    conv = TV.GraphConv(cin, cout, conv="sage", act="leakyrelu", mlp_norm="none").to(DEV)
    ei, w = _graph(n, e, gen)
    ei, w = ei.to(DEV), w.to(DEV)
    x = torch.randn(n, cin, generator=gen).to(DEV)
    mask = torch.rand(n, generator=gen).to(DEV)
    cot = torch.randn(n, cout, generator=gen).to(DEV)

    def run():
        for p in conv.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        out = conv(xg, ei, w, row_scale=mask)
        (out * cot).sum().backward()
        return out.detach(), xg.grad, {k: p.grad.clone() for k, p in conv.named_parameters() if p.grad is not None}

    o1, gx1, gp1 = run()
    o2, gx2, gp2 = run()
    assert torch.equal(o1, o2) and torch.equal(gx1, gx2) and all(torch.equal(gp1[k], gp2[k]) for k in gp1)
    monkeypatch.setattr(TV, "_SAGE_FUSED", False)
    o3, gx3, gp3 = run()
    assert_close(o1, o3, 2e-5, "out", elementwise=True)
    assert_close(gx1, gx3, 2e-5, "grad x", elementwise=True)
    assert set(gp1) == set(gp3)
    for k in gp1:
        assert_close(gp1[k], gp3[k], 2e-5, "grad " + k)


def test_node_embedding_rows():
    from mlgnn import sage as S
    gen = torch.Generator().manual_seed(3)
    nodes, C, B = 15405, 32, 5
    x = torch.rand(B * nodes, 1, generator=gen).to(DEV)
    emb = torch.randn(nodes, C, generator=gen).to(DEV).requires_grad_(True)
    assert S.node_embed_supported(x, emb)
    h = S.node_embed(x, emb)
    ref = (x.reshape(-1, nodes, 1) * emb).reshape(-1, C)
    assert torch.equal(h, ref.detach())
    cot = torch.randn(B * nodes, C, generator=gen).to(DEV)
    (h * cot).sum().backward()
    got = emb.grad.clone()
    emb.grad = None
    (ref * cot).sum().backward()
    assert_close(got, emb.grad, 1e-6, "grad embedding", elementwise=True)
    from mlgnn.ops import row_max_of
    assert torch.equal(row_max_of(h), ref.detach().abs().amax(1))


@pytest.mark.parametrize("B,C,H,W", [(64, 64, 146, 9), (3, 32, 36, 3), (2, 7, 5, 1), (1, 65, 130, 2)])
def test_flatten_of_a_channel_last_tensor(B, C, H, W):
    """torch.flatten(x, start_dim=1) of a [B, C, H, W] tensor that lives channel-last (the head of MultilevelGNN,
    multilevel_gnn.py:277): the tiled transpose gives the same bytes and routes the gradient back channel-last."""
    from mlgnn import sage as S
    gen = torch.Generator().manual_seed(B + C)
    rows = torch.randn(B, H, W, C, generator=gen).to(DEV).requires_grad_(True)
    x = rows.permute(0, 3, 1, 2)                            # logical [B, C, H, W], channel-last in memory
    got = S.flatten_channel_last(x)
    assert got.grad_fn is not None and "TransposeBatched" in type(got.grad_fn).__name__ + str(got.grad_fn.next_functions)
    assert torch.equal(got, torch.flatten(x, start_dim=1))
    cot = torch.randn(B, C * H * W, generator=gen).to(DEV)
    g1, = torch.autograd.grad((got * cot).sum(), rows)
    g2, = torch.autograd.grad((torch.flatten(x, start_dim=1) * cot).sum(), rows)
    assert torch.equal(g1, g2)
    y = torch.randn(B, C, H, W, generator=gen).to(DEV)       # a contiguous tensor takes the plain flatten
    assert torch.equal(S.flatten_channel_last(y), y.reshape(B, -1))
