"""DiffPool contraction for large pooled graphs (csrc/diffpool_large.hip, csrc/gemm_nt.hip; BASELINE configs[4]:
4096 nodes, 1024 clusters, 256 channels, bf16) against the fp64 oracle ``oracle.primitives.dense_diff_pool``
(reference: torch_geometric dense_diff_pool as called from models/diff_pooling.py:59-65).

Two levels.  (i) Kernel arithmetic: given the bf16-rounded softmax S~ the kernel itself produced, every product must
equal the fp64 product of the same rounded operands up to fp32 summation error (1e-5 of the absolute-value bound) --
bf16 operands multiply exactly in fp32.  (ii) End to end against the oracle on unrounded fp64 arithmetic with the bf16
bound: each of the chain's three roundings (S~, T = A S~, the output) is 2^-9 relative, worst case they add, so
|err| <= 2^-7 * (product of absolute values)."""
import math

import pytest
import torch

from oracle import primitives as OP

pytestmark = pytest.mark.gpu

# K = 512 / 1024 / 2048 take the register-resident softmax of the forward prologue, other widths the generic one
SIZES = [(256, 128, 128), (512, 256, 128), (1024, 384, 256), (512, 512, 128), (256, 2048, 128), (384, 1024, 128)]


def _inputs(N, K, C, seed, symmetric=False):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(N, C, generator=g).bfloat16()
    a = torch.rand(N, N, generator=g)
    if symmetric:
        a = 0.5 * (a + a.t())
    a = (a + torch.eye(N)).bfloat16()
    s = (torch.randn(N, K, generator=g) * 2.0).bfloat16()
    return z, a, s


def _align(x):
    return (x + 255) // 256 * 256


def _raw_forward(z, a, s, out_dtype=torch.float32):
    """The C entry point with fp32 outputs (the module surface ties the output dtype to the input's)."""
    from mlgnn import _lib
    N, C = z.shape
    K = s.shape[1]
    dev = z.device
    S = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
    x = torch.empty((K, C), dtype=out_dtype, device=dev)
    ao = torch.empty((K, K), dtype=out_dtype, device=dev)
    stats = torch.empty(3, device=dev)
    scal = torch.empty(2, dtype=out_dtype, device=dev)
    ws = torch.empty(int(_lib.lib.mlgnn_diffpool_large_workspace_bytes(N, K, C)), dtype=torch.uint8, device=dev)
    rc = _lib.lib.mlgnn_diffpool_large_fwd(z.data_ptr(), a.data_ptr(), s.data_ptr(), 1, S.data_ptr(), x.data_ptr(),
                                           ao.data_ptr(), scal.data_ptr(), 0 if out_dtype == torch.float32 else 1, stats.data_ptr(),
                                           ws.data_ptr(), ws.numel(), N, K, C, 1, 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mlgnn_diffpool_large_fwd")
    off_t = _align((2 * K + C) * N * 2)
    stack = ws[:(2 * K + C) * N * 2].view(torch.bfloat16).view(2 * K + C, N)
    T = ws[off_t:off_t + N * K * 2].view(torch.bfloat16).view(N, K)
    G = ws[off_t + _align(N * K * 2):off_t + _align(N * K * 2) + K * K * 2].view(torch.bfloat16).view(K, K)
    return S, x, ao, stats, stack, T, G


@pytest.mark.parametrize("N,K,C", SIZES)
def test_kernel_arithmetic_given_rounded_softmax(N, K, C):
    z, a, s = (t.cuda() for t in _inputs(N, K, C, 1))
    S, x, ao, stats, stack, T, G = _raw_forward(z, a, s)
    soft = torch.softmax(s.double(), -1)
    # S~ is the correctly rounded softmax (fast exp: 2 ulp of fp32 before rounding -> at most one bf16 ulp apart)
    assert ((S.double() - soft).abs() <= 2.0 ** -8 * soft + 1e-30).all()
    Sd, zd, ad = S.double(), z.double(), a.double()
    # stacked transposes
    assert torch.equal(stack[K:2 * K], S.t())
    assert torch.equal(stack[2 * K:], z.t())
    # T = A S~ (fp32 accumulation, one rounding) and its transposed copy
    Tref = ad @ Sd
    assert ((T.double() - Tref).abs() <= 2.0 ** -8 * Tref.abs() + 1e-5 * (ad.abs() @ Sd.abs())).all()
    assert torch.equal(stack[:K], T.t())
    Td = T.double()
    assert ((x.double() - Sd.t() @ zd).abs() <= 1e-5 * (Sd.t() @ zd.abs())).all()
    assert ((ao.double() - Sd.t() @ Td).abs() <= 1e-5 * (Sd.t() @ Td.abs())).all()
    Gref = Sd.t() @ Sd
    assert ((G.double() - Gref).abs() <= 2.0 ** -8 * Gref).all()
    # link from the identity ||A||^2 - 2 <S, A S> + ||S^T S||^2 vs the direct Frobenius norm of the same rounded S
    direct = float(torch.linalg.norm(ad - Sd @ Sd.t()))
    assert abs(float(stats[2]) - direct) <= 1e-4 * direct
    assert abs(float(stats[0]) - direct / (N * N)) <= 1e-4 * direct / (N * N)
    ent = float((-soft * torch.log(soft + 1e-15)).sum(-1).mean())
    assert abs(float(stats[1]) - ent) <= 1e-4 * abs(ent)


@pytest.mark.parametrize("N,K,C", SIZES)
def test_forward_backward_vs_oracle(N, K, C):
    from mlgnn.dense import dense_diff_pool
    z, a, s = _inputs(N, K, C, 2)
    zd, ad, sd = z.double().requires_grad_(True), a.double(), s.double().requires_grad_(True)
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    g = torch.Generator().manual_seed(3)
    wx, wa = torch.randn(K, C, generator=g).double(), torch.randn(K, K, generator=g).double() / K
    (rx[0] * wx).sum().add((ra[0] * wa).sum()).add(rl * 3e4).add(re * 2.0).backward()

    zc, sc = z.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    x, ao, link, ent = dense_diff_pool(zc, a.cuda(), sc)
    assert x.dtype == torch.bfloat16 and x.shape == (1, K, C) and ao.shape == (1, K, K)
    soft = torch.softmax(sd.detach(), -1)
    bound_x = soft.t() @ zd.detach().abs()
    bound_a = soft.t() @ ad.abs() @ soft
    # worst case per factor: S~ is within ONE bf16 ulp of the exact softmax (2^-8: the fast exponential may flip the
    # rounding, see test_kernel_arithmetic_given_rounded_softmax), T and the output are correctly rounded (2^-9 each).
    # X' = S~^T Z: 2^-8 + 2^-9 < 2^-7.  A' = S~^T (A S~) carries S~ twice: 2 * 2^-8 + 2 * 2^-9 = 3 * 2^-8 -- reached
    # when a column of S is dominated by one node (few nodes, many clusters: the 256 x 2048 case), where the errors of
    # the dominant entries do not average out.
    tol = 2.0 ** -7
    assert ((x[0].double().cpu() - rx[0].detach()).abs() <= tol * bound_x).all()
    assert ((ao[0].double().cpu() - ra[0].detach()).abs() <= 3 * 2.0 ** -8 * bound_a).all()
    assert abs(float(link) - float(rl)) <= 2.0 ** -7 * float(rl)
    assert abs(float(ent) - float(re)) <= 2.0 ** -7 * abs(float(re))
    ((x[0].float() * wx.float().cuda()).sum() + (ao[0].float() * wa.float().cuda()).sum() + link.float() * 3e4
     + ent.float() * 2.0).backward()
    for got, ref, name in ((zc.grad, zd.grad, "grad z"), (sc.grad, sd.grad, "grad logits")):
        err = float(torch.linalg.norm(got.double().cpu() - ref)) / float(torch.linalg.norm(ref))
        assert err <= 2.0 ** -6, (name, err)             # bf16 operands and bf16 gradient storage: norm-wise 1.6 %


def test_symmetric_adjacency_shortcut_is_bitwise_identical():
    from mlgnn.dense import dense_diff_pool
    N, K, C = 512, 256, 128
    z, a, s = _inputs(N, K, C, 4, symmetric=True)
    assert torch.equal(a, a.t())
    outs = []
    for sym in (False, True):
        zc, sc = z.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
        x, ao, link, ent = dense_diff_pool(zc, a.cuda(), sc, adj_symmetric=sym)
        (x.float().sum() + (ao.float() ** 2).sum() + 1e4 * link.float() + ent.float()).backward()
        outs.append((x, ao, link, ent, zc.grad, sc.grad))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


def test_batched_and_reproducible():
    from mlgnn.dense import dense_diff_pool
    N, K, C = 256, 128, 128
    zs, ss = [], []
    for b in range(2):
        z, a, s = _inputs(N, K, C, 10 + b)
        zs.append(z)
        ss.append(s)
    z, s, a = torch.stack(zs).cuda(), torch.stack(ss).cuda(), a.cuda()
    x1, a1, l1, e1 = dense_diff_pool(z, a, s)
    x2, a2, l2, e2 = dense_diff_pool(z, a, s)
    assert torch.equal(x1, x2) and torch.equal(a1, a2) and torch.equal(l1, l2) and torch.equal(e1, e2)
    rx, ra, rl, re = OP.dense_diff_pool(z.double().cpu(), a.double().cpu(), s.double().cpu())
    assert x1.shape == rx.shape and a1.shape == ra.shape
    assert abs(float(l1) - float(rl)) <= 2.0 ** -7 * float(rl)
    assert abs(float(e1) - float(re)) <= 2.0 ** -7 * abs(float(re))
    assert float((x1.double().cpu() - rx).abs().max()) <= 2.0 ** -6 * float(rx.abs().max())


@pytest.mark.parametrize("shared_adj", [False, True])
def test_batch_is_one_grouped_launch_with_the_references_batch_semantics(shared_adj):
    """B = 3 pooled graphs through the grouped launches (grid.y = graph): every graph's outputs are bitwise those of a
    call on that graph alone, the scalars are the batch's (ONE Frobenius norm over the batch / numel of the adj ARGUMENT,
    entropy over all nodes) and every gradient -- z, logits, and the adjacency (summed over the batch when it is shared)
    -- matches the fp64 oracle on the batched call norm-wise within the bf16 bound."""
    from mlgnn.dense import dense_diff_pool
    N, K, C, B = 256, 128, 128, 3
    zs, as_, ss = zip(*[_inputs(N, K, C, 40 + b) for b in range(B)])
    z, s = torch.stack(zs), torch.stack(ss)
    a = as_[0][None] if shared_adj else torch.stack(as_)
    zd, ad, sd = z.double().requires_grad_(True), a.double().requires_grad_(True), s.double().requires_grad_(True)
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    g = torch.Generator().manual_seed(8)
    wx, wa = torch.randn(B, K, C, generator=g).double(), torch.randn(B, K, K, generator=g).double() / K
    ((rx * wx).sum() + (ra * wa).sum() + rl * 3e4 + re * 2.0).backward()
    zc, ac, sc = z.cuda().requires_grad_(True), a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    x, ao, link, ent = dense_diff_pool(zc, ac, sc)
    assert x.shape == (B, K, C) and ao.shape == (B, K, K)
    for b in range(B):
        xb, ab_, _, _ = dense_diff_pool(z[b:b + 1].cuda(), a[0 if shared_adj else b][None].cuda(), s[b:b + 1].cuda())
        assert torch.equal(x[b], xb[0]) and torch.equal(ao[b], ab_[0])
    assert abs(float(link) - float(rl)) <= 2.0 ** -7 * float(rl)
    assert abs(float(ent) - float(re)) <= 2.0 ** -7 * abs(float(re))
    ((x.float() * wx.float().cuda()).sum() + (ao.float() * wa.float().cuda()).sum() + link.float() * 3e4
     + ent.float() * 2.0).backward()
    assert ac.grad.shape == a.shape
    for got, ref, name in ((ac.grad, ad.grad, "grad adj"), (zc.grad, zd.grad, "grad z"), (sc.grad, sd.grad, "grad logits")):
        err = float(torch.linalg.norm(got.double().cpu() - ref)) / float(torch.linalg.norm(ref))
        assert err <= 2.0 ** -6, (name, err)


def test_configs4_size_properties():
    """BASELINE configs[4] size (4096 nodes, 1024 clusters, 256 channels): too large for the fp64 oracle in seconds on
    all outputs, so the size-independent properties: column sums (S^T Z summed over clusters = column sums of Z since
    softmax rows sum to 1), total edge mass (sum A' = sum_ij A_ij up to rounding), a sampled block of A' against fp64."""
    from mlgnn.dense import dense_diff_pool
    N, K, C = 4096, 1024, 256
    z, a, s = (t.cuda() for t in _inputs(N, K, C, 5))
    zc, sc = z.clone().requires_grad_(True), s.clone().requires_grad_(True)
    x, ao, link, ent = dense_diff_pool(zc, a, sc)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all() and torch.isfinite(ao).all()
    col = x[0].double().sum(0)
    want = z.double().sum(0)
    assert ((col - want).abs() <= 2.0 ** -7 * z.double().abs().sum(0)).all()
    mass, want_mass = float(ao[0].double().sum()), float(a.double().sum())
    assert abs(mass - want_mass) <= 2.0 ** -7 * want_mass
    soft = torch.softmax(s.double(), -1)
    blk = soft[:, :64].t() @ (a.double() @ soft[:, 64:128])
    assert ((ao[0, :64, 64:128].double() - blk).abs() <= 2.0 ** -7 * blk.abs()).all()
    direct = float(torch.linalg.norm(a.double() - soft @ soft.t())) / (N * N)
    assert abs(float(link) - direct) <= 2.0 ** -7 * direct
    assert 0.0 < float(ent) <= math.log(K) + 1e-3
    (x.float().sum() + ao.float().sum() + link.float() + ent.float()).backward()
    assert torch.isfinite(zc.grad).all() and torch.isfinite(sc.grad).all()
    # d/dz of sum(S^T z) is the row sum of S = 1 for every entry
    assert ((zc.grad.float() - 1.0).abs() <= 2.0 ** -6).all()


# ---- fp32 inputs: every product as three bf16 terms on the matrix cores (fp32-level accuracy) ------------------------

@pytest.mark.parametrize("N,K,C", SIZES + [(4096, 1024, 256)])
def test_fp32_path_matches_fp64_oracle_to_1e4(N, K, C):
    """north_star's fp32 bar on the large path: outputs within 1e-4 of the fp64 oracle, elementwise against the
    absolute-value bound of each product (the three-term split leaves 2^-17 per operand), gradients likewise."""
    from mlgnn.dense import dense_diff_pool
    g = torch.Generator().manual_seed(21)
    z = torch.randn(N, C, generator=g)
    a = torch.rand(N, N, generator=g) + torch.eye(N)
    s = torch.randn(N, K, generator=g) * 2.0
    dev = "cuda:0"
    # the oracle itself runs on the GPU here, in fp64 (it is plain torch): 80 GFLOP of fp64 at the full size
    zd, ad, sd = z.double().to(dev).requires_grad_(True), a.double().to(dev), s.double().to(dev).requires_grad_(True)
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    wx, wa = torch.randn(K, C, generator=g).double().to(dev), (torch.randn(K, K, generator=g) / K).double().to(dev)
    ((rx[0] * wx).sum() + (ra[0] * wa).sum() + rl * 3e4 + re * 2.0).backward()
    zc, sc = z.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    x, ao, link, ent = dense_diff_pool(zc, a.to(dev), sc)
    assert x.dtype == torch.float32 and x.shape == (1, K, C)
    soft = torch.softmax(sd.detach(), -1)
    bound_x = soft.t() @ zd.detach().abs()
    bound_a = soft.t() @ ad @ soft
    assert ((x[0].double() - rx[0].detach()).abs() <= 1e-4 * bound_x).all()
    assert ((ao[0].double() - ra[0].detach()).abs() <= 1e-4 * bound_a).all()
    assert abs(float(link) - float(rl)) <= 1e-4 * float(rl)
    assert abs(float(ent) - float(re)) <= 1e-4 * abs(float(re))
    ((x[0] * wx.float()).sum() + (ao[0] * wa.float()).sum() + link * 3e4 + ent * 2.0).backward()
    for got, ref, name in ((zc.grad, zd.grad, "grad z"), (sc.grad, sd.grad, "grad logits")):
        err = float((got.double() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (name, err)


# ---- adjacency gradient: the second pooling level's adjacency is the first level's S^T A S ---------------------------

@pytest.mark.parametrize("N,K,C", SIZES)
def test_adjacency_gradient_bf16(N, K, C):
    """dA = S (dA' - cI) S^T + c A against autograd of the fp64 oracle (models/diff_pooling.py:116-127 feeds A' of
    level i to level i+1 as its adjacency): norm-wise within the bf16 bound, like the other gradients."""
    from mlgnn.dense import dense_diff_pool
    z, a, s = _inputs(N, K, C, 7)
    zd, ad, sd = z.double().requires_grad_(True), a.double().requires_grad_(True), s.double().requires_grad_(True)
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    g = torch.Generator().manual_seed(8)
    wx, wa = torch.randn(K, C, generator=g).double(), torch.randn(K, K, generator=g).double() / K
    (rx[0] * wx).sum().add((ra[0] * wa).sum()).add(rl * 3e4).add(re * 2.0).backward()
    zc, ac, sc = z.cuda().requires_grad_(True), a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    x, ao, link, ent = dense_diff_pool(zc, ac, sc)
    ((x[0].float() * wx.float().cuda()).sum() + (ao[0].float() * wa.float().cuda()).sum() + link.float() * 3e4
     + ent.float() * 2.0).backward()
    assert ac.grad is not None and ac.grad.shape == (N, N) and ac.grad.dtype == torch.bfloat16
    for got, ref, name in ((ac.grad, ad.grad, "grad adj"), (zc.grad, zd.grad, "grad z"), (sc.grad, sd.grad, "grad logits")):
        err = float(torch.linalg.norm(got.double().cpu() - ref)) / float(torch.linalg.norm(ref))
        assert err <= 2.0 ** -6, (name, err)


@pytest.mark.parametrize("N,K,C", SIZES)
def test_adjacency_gradient_fp32(N, K, C):
    from mlgnn.dense import dense_diff_pool
    g = torch.Generator().manual_seed(31)
    dev = "cuda:0"
    z = torch.randn(N, C, generator=g)
    a = torch.rand(N, N, generator=g) + torch.eye(N)
    s = torch.randn(N, K, generator=g) * 2.0
    zd, ad, sd = (t.double().to(dev).requires_grad_(True) for t in (z, a, s))
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    wx, wa = torch.randn(K, C, generator=g).double().to(dev), (torch.randn(K, K, generator=g) / K).double().to(dev)
    ((rx[0] * wx).sum() + (ra[0] * wa).sum() + rl * 3e4 + re * 2.0).backward()
    zc, ac, sc = (t.to(dev).requires_grad_(True) for t in (z, a, s))
    x, ao, link, ent = dense_diff_pool(zc, ac, sc)
    ((x[0] * wx.float()).sum() + (ao[0] * wa.float()).sum() + link * 3e4 + ent * 2.0).backward()
    err = float((ac.grad.double() - ad.grad).abs().max())
    assert err <= 1e-4 * max(1.0, float(ad.grad.abs().max())), err


def test_two_level_chain_gradients_through_both_levels():
    """4096 -> 1024 -> 256 (BASELINE configs[4]): level 2 takes level 1's (x', A') -- both levels stay on the
    matrix-core chain, the gradient of level 1's inputs passes through level 2's adjacency gradient.  fp32 inputs,
    fp64 oracle on the device, 1e-4 like the single-level fp32 test."""
    from mlgnn.dense import dense_diff_pool
    g = torch.Generator().manual_seed(41)
    dev = "cuda:0"
    N, K1, K2, C = 4096, 1024, 256, 256
    z = torch.randn(N, C, generator=g)
    a = torch.rand(N, N, generator=g) + torch.eye(N)
    s1 = torch.randn(N, K1, generator=g) * 2.0
    s2 = torch.randn(K1, K2, generator=g) * 2.0
    zd, sd1, sd2 = (t.double().to(dev).requires_grad_(True) for t in (z, s1, s2))
    ad = a.double().to(dev)
    x1, a1, l1, e1 = OP.dense_diff_pool(zd, ad, sd1)
    x2, a2, l2, e2 = OP.dense_diff_pool(x1[0], a1[0], sd2)
    wx, wa = torch.randn(K2, C, generator=g).double().to(dev), (torch.randn(K2, K2, generator=g) / K2).double().to(dev)
    ((x2[0] * wx).sum() + (a2[0] * wa).sum() + (l1 + l2) * 3e4 + (e1 + e2) * 2.0).backward()
    zc, sc1, sc2 = (t.to(dev).requires_grad_(True) for t in (z, s1, s2))
    y1, b1, m1, f1 = dense_diff_pool(zc, a.to(dev), sc1)
    assert b1.requires_grad
    y2, b2, m2, f2 = dense_diff_pool(y1[0], b1[0], sc2)
    ((y2[0] * wx.float()).sum() + (b2[0] * wa.float()).sum() + (m1 + m2) * 3e4 + (f1 + f2) * 2.0).backward()
    for got, ref, name in ((y2[0], x2[0].detach(), "x''"), (b2[0], a2[0].detach(), "A''")):
        err = float((got.detach().double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err <= 1e-4, (name, err)
    for got, ref, name in ((zc.grad, zd.grad, "grad z"), (sc1.grad, sd1.grad, "grad logits 1"), (sc2.grad, sd2.grad, "grad logits 2")):
        err = float((got.double() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (name, err)


@pytest.mark.parametrize("shared_adj", [False, True])
def test_fp32_batch_is_grouped_launches_with_the_references_batch_semantics(shared_adj):
    """fp32, B = 3 through ``mlgnn_diffpool_large_f32_fwd`` / ``_bwd`` (grid.y = graph): per-graph outputs bitwise those of
    single calls, the batch's scalars and every gradient (z, logits, adjacency -- summed over the batch when shared)
    within 1e-4 of the fp64 oracle on the batched call."""
    from mlgnn.dense import dense_diff_pool
    N, K, C, B = 256, 128, 128, 3
    g = torch.Generator().manual_seed(51)
    dev = "cuda:0"
    z = torch.randn(B, N, C, generator=g)
    a = torch.rand(1 if shared_adj else B, N, N, generator=g) + torch.eye(N)
    s = torch.randn(B, N, K, generator=g) * 2.0
    zd, ad, sd = (t.double().to(dev).requires_grad_(True) for t in (z, a, s))
    rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
    wx, wa = torch.randn(B, K, C, generator=g).double().to(dev), (torch.randn(B, K, K, generator=g) / K).double().to(dev)
    ((rx * wx).sum() + (ra * wa).sum() + rl * 3e4 + re * 2.0).backward()
    zc, ac, sc = (t.to(dev).requires_grad_(True) for t in (z, a, s))
    x, ao, link, ent = dense_diff_pool(zc, ac, sc)
    assert x.shape == (B, K, C) and ao.shape == (B, K, K) and x.dtype == torch.float32
    for b in range(B):
        xb, ab_, _, _ = dense_diff_pool(z[b:b + 1].to(dev), a[0 if shared_adj else b][None].to(dev), s[b:b + 1].to(dev))
        assert torch.equal(x[b], xb[0]) and torch.equal(ao[b], ab_[0])
    assert abs(float(link) - float(rl)) <= 1e-4 * float(rl)
    assert abs(float(ent) - float(re)) <= 1e-4 * abs(float(re))
    assert float((x.double() - rx.detach()).abs().max()) <= 1e-4 * max(1.0, float(rx.abs().max()))
    assert float((ao.double() - ra.detach()).abs().max()) <= 1e-4 * max(1.0, float(ra.abs().max()))
    ((x * wx.float()).sum() + (ao * wa.float()).sum() + link * 3e4 + ent * 2.0).backward()
    assert ac.grad.shape == a.shape
    for got, ref, name in ((ac.grad, ad.grad, "grad adj"), (zc.grad, zd.grad, "grad z"), (sc.grad, sd.grad, "grad logits")):
        err = float((got.double() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (name, err)


def test_fp32_symmetric_shortcut_and_argument_errors():
    """A symmetric adjacency with ``adj_symmetric=True`` (T2 = T: no A^T product) gives the gradients of the general
    path to rounding; the C entry point refuses bad shapes / workspaces / pointers with its error codes."""
    from mlgnn import _lib
    from mlgnn.dense import dense_diff_pool
    N, K, C = 256, 128, 128
    g = torch.Generator().manual_seed(61)
    z, s = torch.randn(N, C, generator=g).cuda(), (torch.randn(N, K, generator=g) * 2.0).cuda()
    a = torch.rand(N, N, generator=g)
    a = ((a + a.t()) * 0.5).cuda()
    grads = []
    for sym in (False, True):
        zc, sc = z.clone().requires_grad_(True), s.clone().requires_grad_(True)
        x, ao, link, ent = dense_diff_pool(zc, a, sc, adj_symmetric=sym)
        (x.sum() + (ao ** 2).sum() + 1e4 * link + ent).backward()
        grads.append((zc.grad, sc.grad))
    for u, v in zip(*grads):
        assert float((u - v).abs().max()) <= 1e-5 * max(1.0, float(u.abs().max()))
    L = _lib.lib
    assert L.mlgnn_diffpool_large_f32_workspace_bytes(100, 128, 128) == -2
    need = int(L.mlgnn_diffpool_large_f32_workspace_bytes(N, K, C))
    assert 0 < int(L.mlgnn_diffpool_large_f32_saved_bytes(N, K, C)) < need
    S, xo, aout = torch.empty(N, K).cuda(), torch.empty(K, C).cuda(), torch.empty(K, K).cuda()
    scal, stats = torch.empty(2).cuda(), torch.empty(3).cuda()
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = [z.data_ptr(), a.data_ptr(), s.data_ptr(), S.data_ptr(), xo.data_ptr(), aout.data_ptr(), scal.data_ptr(),
            stats.data_ptr(), ws.data_ptr()]
    assert L.mlgnn_diffpool_large_f32_fwd(*args, need - 1, N, K, C, 1, 0, st) == -5
    assert L.mlgnn_diffpool_large_f32_fwd(*args, need, N, K + 1, C, 1, 0, st) == -2
    assert L.mlgnn_diffpool_large_f32_fwd(*args[:3], None, *args[4:], need, N, K, C, 1, 0, st) == -1
    assert L.mlgnn_diffpool_large_f32_fwd(*args, need, N, K, C, 1, 0, st) == 0
    torch.cuda.synchronize()
    assert float((S.sum(-1) - 1.0).abs().max()) <= 1e-5
