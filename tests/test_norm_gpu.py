"""Fused LayerNorm(+ReLU) kernels vs torch.nn.functional on the CPU (fp32, 1e-4)."""
import pytest
import torch
import torch.nn.functional as F

from _util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,d", [(1, 4), (7, 8), (1000, 32), (513, 64), (2049, 100), (4097, 128), (3000, 256),
                                    (70000, 128), (2000, 264), (3001, 384), (5000, 512)])
@pytest.mark.parametrize("relu", [False, True])
def test_layer_norm_act(rows, d, relu):
    from mlgnn.norm import layer_norm_act
    gen = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=gen) * 2 + 0.5).requires_grad_(True)
    w = (torch.rand(d, generator=gen) + 0.5).requires_grad_(True)
    b = (torch.randn(d, generator=gen) * 0.3).requires_grad_(True)
    cot = torch.randn(rows, d, generator=gen)
    ref = F.layer_norm(x, (d,), w, b, 1e-5)
    ref = F.relu(ref) if relu else ref
    gr = torch.autograd.grad((ref * cot).sum(), [x, w, b])
    dev = "cuda:0"
    xd, wd, bd = (t.detach().to(dev).requires_grad_(True) for t in (x, w, b))
    out = layer_norm_act(xd, wd, bd, 1e-5, relu)
    assert_close(out, ref, 1e-4, "ln fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, wd, bd])
    for name, g, r in zip(("x", "gamma", "beta"), got, gr):
        assert_close(g, r, 1e-4, "ln grad " + name)


@pytest.mark.parametrize("rows,d,dtype", [(1000, 128, torch.float32), (4097, 256, torch.float32), (513, 36, torch.float32),
                                          (3000, 512, torch.float32), (2000, 256, torch.bfloat16)])
@pytest.mark.parametrize("fork", [False, True])
def test_layer_norm_act_with_fused_dropout(rows, d, dtype, fork):
    """norm -> ReLU -> dropout in one pass each way: with an explicit keep mask the result must equal the composition
    ``relu(LN(x)) * mask / (1 - p)`` and its gradients; with a drawn mask the kept fraction must be 1 - p."""
    from mlgnn.norm import layer_norm_act, layer_norm_act_fork
    gen = torch.Generator().manual_seed(rows + d)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    rb = (lambda t: t) if dtype == torch.float32 else (lambda t: t.to(torch.bfloat16).float())
    p = 0.3
    x = rb(torch.randn(rows, d, generator=gen) * 2 + 0.5).requires_grad_(True)
    w = rb(torch.rand(d, generator=gen) + 0.5).requires_grad_(True)
    b = rb(torch.randn(d, generator=gen) * 0.3).requires_grad_(True)
    cot, cot2 = rb(torch.randn(rows, d, generator=gen)), rb(torch.randn(rows, d, generator=gen))
    mask = (torch.rand(rows, d, generator=gen) > p).to(torch.uint8)
    pre = F.layer_norm(x, (d,), w, b, 1e-5)
    ref = F.relu(pre) * mask / (1 - p)
    gr = torch.autograd.grad((ref * cot).sum() + ((x * cot2).sum() if fork else 0), [x, w, b])
    keep = pre.detach().abs() > 1e-4                         # (ReLU mask ties, see the bf16 LayerNorm test)
    dev = "cuda:0"
    xd, wd, bd = (t.detach().to(dev).to(dtype).requires_grad_(True) for t in (x, w, b))
    if fork:
        out, ident = layer_norm_act_fork(xd, wd, bd, 1e-5, True, dropout_p=p, dropout_mask=mask.to(dev))
        loss = (out.float() * cot.to(dev)).sum() + (ident.float() * cot2.to(dev)).sum()
    else:
        out = layer_norm_act(xd, wd, bd, 1e-5, True, dropout_p=p, dropout_mask=mask.to(dev))
        loss = (out.float() * cot.to(dev)).sum()
    assert_close(out.float(), ref, tol, "ln + dropout fwd")
    assert bool((out[(mask == 0).to(dev)] == 0).all())
    got = torch.autograd.grad(loss, [xd, wd, bd])
    assert_close(got[0].float().cpu() * keep, gr[0] * keep, tol, "grad x")
    assert_close(got[1].float(), gr[1], tol * 2, "grad gamma")
    assert_close(got[2].float(), gr[2], tol * 2, "grad beta")
    drawn = layer_norm_act(torch.ones(rows, d, device=dev, dtype=dtype) + xd.detach(), wd.detach() * 0, bd.detach() * 0 + 1,
                           1e-5, False, dropout_p=p)                        # LN output == beta == 1 everywhere
    frac = float((drawn != 0).float().mean())
    assert abs(frac - (1 - p)) < 0.02 and abs(float(drawn.float().max()) - 1 / (1 - p)) < 2e-2
    assert torch.equal(layer_norm_act(xd, wd, bd, 1e-5, True, dropout_p=0.0), layer_norm_act(xd, wd, bd, 1e-5, True))


def test_unsupported_width_uses_aten_on_device():
    from mlgnn.norm import layer_norm_act
    for d in (258, 260, 520):                                   # not a multiple of 4 / of 8 beyond 256 / too wide
        x = torch.randn(10, d, device="cuda:0")
        w, b = torch.ones(d, device="cuda:0"), torch.zeros(d, device="cuda:0")
        assert_close(layer_norm_act(x, w, b, 1e-5, True), F.relu(F.layer_norm(x, (d,), w, b)), 1e-6)


@pytest.mark.parametrize("rows,d", [(1, 4), (1000, 32), (777, 100), (5000, 128), (3000, 256)])
def test_msg_norm_add(rows, d):
    from mlgnn.norm import msg_norm_add
    gen = torch.Generator().manual_seed(rows * 3 + d)
    x = torch.randn(rows, d, generator=gen, requires_grad=True)
    m = (torch.rand(rows, d, generator=gen) * 3).requires_grad_(True)
    with torch.no_grad():
        m[0] = 0.0                                     # zero message row: F.normalize clamps the norm
        if rows > 2:
            x[2] = 0.0                                 # zero feature row: ||x|| has a zero sub-gradient
    s = torch.tensor([0.7], requires_grad=True)
    cot = torch.randn(rows, d, generator=gen)
    ref = x + F.normalize(m, p=2.0, dim=1) * x.norm(p=2, dim=1, keepdim=True) * s
    gr = torch.autograd.grad((ref * cot).sum(), [x, m, s])
    dev = "cuda:0"
    xd, md, sd = (t.detach().to(dev).requires_grad_(True) for t in (x, m, s))
    out = msg_norm_add(xd, md, sd)
    assert_close(out, ref, 1e-4, "msgnorm fwd")
    got = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, md, sd])
    for name, g, r in zip(("x", "m", "scale"), got, gr):
        assert_close(g, r, 1e-4, "msgnorm grad " + name)


@pytest.mark.parametrize("rows,d", [(1000, 128), (8193, 256), (37, 12)])
def test_residual_block_through_fork_and_gemm_epilogue(rows, d):
    """h' = Linear(relu(LN(h))) + h with the add in the GEMM epilogue and its gradient inside the LayerNorm
    backward kernel equals the plain composition (values and every gradient)."""
    import torch.nn.functional as F
    from mlgnn.dense import linear
    from mlgnn.norm import layer_norm_act_fork
    gen = torch.Generator().manual_seed(rows)
    h = torch.randn(rows, d, generator=gen, requires_grad=True)
    g = (torch.rand(d, generator=gen) + 0.5).requires_grad_(True)
    b = torch.randn(d, generator=gen).requires_grad_(True)
    w = (torch.randn(d, d, generator=gen) * 0.1).requires_grad_(True)
    wb = torch.randn(d, generator=gen).requires_grad_(True)
    cot = torch.randn(rows, d, generator=gen)
    leaves = [h, g, b, w, wb]
    ref = F.linear(torch.relu(F.layer_norm(h, (d,), g, b, 1e-5)), w, wb) + h
    gr = torch.autograd.grad((ref * cot).sum(), leaves)
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    y, identity = layer_norm_act_fork(dl[0], dl[1], dl[2], 1e-5, relu=True)
    out = linear(y, dl[3], dl[4], residual=identity)
    assert_close(out, ref, 1e-4, "residual block fwd")
    got = torch.autograd.grad((out * cot.cuda()).sum(), dl)
    for name, a, r in zip(("h", "gamma", "beta", "W", "bias"), got, gr):
        assert_close(a, r, 1e-4, "residual block grad " + name)
    # the identity output alone (norm branch unused) still carries its gradient
    y2, id2 = layer_norm_act_fork(dl[0], dl[1], dl[2], 1e-5, relu=True)
    (gh,) = torch.autograd.grad((id2 * cot.cuda()).sum() + 0.0 * y2.sum(), [dl[0]])
    assert_close(gh, cot, 1e-5, "identity branch only")


@pytest.mark.parametrize("rows,k,hdim,o,res", [(9000, 128, 256, 128, True), (8200, 64, 128, 64, False),
                                             (20011, 128, 256, 128, False), (8192, 64, 256, 128, True),
                                             (8300, 128, 128, 256, True)])
def test_fused_mlp2_matches_the_unfused_composition(rows, k, hdim, o, res):
    """Linear -> LayerNorm -> ReLU -> Linear (+ residual) with the normalised hidden activation stored once
    (first GEMM's epilogue) and the affine map + ReLU applied by its consumers: values and every gradient."""
    from mlgnn.dense import fused_mlp2, fused_mlp2_supported
    gen = torch.Generator().manual_seed(rows + hdim)
    x = torch.randn(rows, k, generator=gen, requires_grad=True)
    w1 = (torch.randn(hdim, k, generator=gen) * 0.2).requires_grad_(True)
    b1 = torch.randn(hdim, generator=gen).requires_grad_(True)
    g = (torch.rand(hdim, generator=gen) + 0.5).requires_grad_(True)
    be = (torch.randn(hdim, generator=gen) * 0.3).requires_grad_(True)
    w2 = (torch.randn(o, hdim, generator=gen) * 0.1).requires_grad_(True)
    b2 = torch.randn(o, generator=gen).requires_grad_(True)
    r = torch.randn(rows, o, generator=gen).requires_grad_(True) if res else None
    cot = torch.randn(rows, o, generator=gen)
    leaves = [x, w1, b1, g, be, w2, b2] + ([r] if res else [])
    hid = torch.relu(F.layer_norm(F.linear(x, w1, b1), (hdim,), g, be, 1e-5))
    ref = F.linear(hid, w2, b2) + (r if res else 0)
    gr = torch.autograd.grad((ref * cot).sum(), leaves)
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    assert fused_mlp2_supported(dl[0], dl[1], dl[5])
    out = fused_mlp2(dl[0], dl[1], dl[2], dl[3], dl[4], 1e-5, dl[5], dl[6], dl[7] if res else None)
    assert_close(out, ref, 1e-4, "fused mlp fwd")
    got = torch.autograd.grad((out * cot.cuda()).sum(), dl)
    names = ["x", "W1", "b1", "gamma", "beta", "W2", "b2"] + (["residual"] if res else [])
    for name, a, b in zip(names, got, gr):
        assert_close(a, b, 1e-4, "fused mlp grad " + name)


@pytest.mark.parametrize("rows,k,hdim,o,res,relu,use", [(9000, 128, 256, 128, True, True, "both"),
                                                       (8200, 64, 128, 64, False, True, "both"),
                                                       (20011, 128, 256, 128, True, False, "y"),
                                                       (8192, 64, 256, 128, True, True, "out"),
                                                       (8300, 128, 128, 64, False, False, "both")])
def test_fused_mlp2_with_the_next_norm_in_its_epilogue(rows, k, hdim, o, res, relu, use):
    """``(out, relu?(LayerNorm(out)))`` from the second GEMM's epilogue (csrc/tallgemm.hip POST): both values and every
    gradient -- with cotangents on both outputs (a middle res+ block: identity branch + next conv), on ``y`` alone (the
    final norm after the last conv) and on ``out`` alone -- against the unfused composition on the CPU."""
    from mlgnn.dense import fused_mlp2, fused_mlp2_post_supported
    gen = torch.Generator().manual_seed(rows + hdim + 1)
    x = torch.randn(rows, k, generator=gen, requires_grad=True)
    w1 = (torch.randn(hdim, k, generator=gen) * 0.2).requires_grad_(True)
    b1 = torch.randn(hdim, generator=gen).requires_grad_(True)
    g = (torch.rand(hdim, generator=gen) + 0.5).requires_grad_(True)
    be = (torch.randn(hdim, generator=gen) * 0.3).requires_grad_(True)
    w2 = (torch.randn(o, hdim, generator=gen) * 0.1).requires_grad_(True)
    b2 = torch.randn(o, generator=gen).requires_grad_(True)
    g2 = (torch.rand(o, generator=gen) + 0.5).requires_grad_(True)
    be2 = (torch.randn(o, generator=gen) * 0.3).requires_grad_(True)
    leaves = [x, w1, b1, g, be, w2, b2, g2, be2]
    if res:
        leaves.append((torch.randn(rows, o, generator=gen) * 2.0).requires_grad_(True))
    c_out, c_y = torch.randn(rows, o, generator=gen), torch.randn(rows, o, generator=gen)

    def run(lv, dev):
        if dev == "cpu":
            hid = torch.relu(F.layer_norm(F.linear(lv[0], lv[1], lv[2]), (hdim,), lv[3], lv[4], 1e-5))
            out = F.linear(hid, lv[5], lv[6]) + (lv[9] if res else 0)
            y = F.layer_norm(out, (o,), lv[7], lv[8], 1e-5)
            y = torch.relu(y) if relu else y
        else:
            assert fused_mlp2_post_supported(lv[0], lv[1], lv[5], lv[7])
            out, y = fused_mlp2(lv[0], lv[1], lv[2], lv[3], lv[4], 1e-5, lv[5], lv[6], lv[9] if res else None,
                                (lv[7], lv[8], 1e-5, relu))
        loss = 0.0
        if use in ("both", "out"):
            loss = loss + (out * c_out.to(dev)).sum()
        if use in ("both", "y"):
            loss = loss + (y * c_y.to(dev)).sum()
        return out, y, torch.autograd.grad(loss, lv, allow_unused=True)

    out_r, y_r, gr = run(leaves, "cpu")
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    out, y, got = run(dl, "cuda")
    assert_close(out, out_r, 1e-4, "out", elementwise=True)
    assert_close(y, y_r, 1e-4, "y", elementwise=True)
    names = ["x", "W1", "b1", "gamma", "beta", "W2", "b2", "post gamma", "post beta"] + (["residual"] if res else [])
    for name, a, b in zip(names, got, gr):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, name
        else:
            assert_close(a, b, 1e-4, "grad " + name)
