"""Projection-pooling kernels vs the oracle's literal [B,G,C,k] scatter_reduce form."""
import pytest
import torch

from _util import assert_close
from oracle import models as M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,K", [(1, 1), (8, 2), (32, 3), (128, 2), (100, 4), (256, 2)])
@pytest.mark.parametrize("match_mask", [True, False])
@pytest.mark.parametrize("sorted_segments", [True, False])
def test_segment_project_matches_oracle(C, K, match_mask, sorted_segments):
    from mlgnn.project import segment_project
    gen = torch.Generator().manual_seed(C * 10 + K)
    B, NN, G, S = 3, 50, 700, 438
    x = torch.randn(B * NN, C, generator=gen, requires_grad=True)
    w = (torch.randn(G, K, generator=gen) * 0.3).requires_grad_(True)
    match = torch.randint(0, NN, (B, G), generator=gen)
    match[:, ::9] = -1
    match[1, 5] = -3                                  # wraps to another graph's row when unmasked
    seg = torch.randint(0, S, (B, G), generator=gen)
    seg[:, :80] = 7                                   # one long segment (> one 64-member chunk)
    if sorted_segments:
        seg = torch.sort(seg, dim=1)[0]
    cot = torch.randn(B, C, S, K, generator=gen)
    ref = M.projection_pool(x, match, seg, w, None, NN, S, match_mask)
    gx_ref, gw_ref = torch.autograd.grad((ref * cot).sum(), [x, w])

    dev = "cuda:0"
    xd, wd = x.detach().to(dev).requires_grad_(True), w.detach().to(dev).requires_grad_(True)
    out = segment_project(xd, match.to(dev), seg.to(dev), wd, NN, S, match_mask)
    assert tuple(out.shape) == (B, C, S, K)
    assert_close(out, ref, 1e-4, "projection fwd")
    gx, gw = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, wd])
    assert_close(gx, gx_ref, 1e-4, "projection grad x")
    assert_close(gw, gw_ref, 1e-4, "projection grad w")


def test_membership_tables_are_cached_by_identity():
    from mlgnn.project import membership_tables
    dev = "cuda:0"
    match = torch.randint(0, 10, (2, 30), device=dev)
    seg = torch.randint(0, 5, (2, 30), device=dev)
    a = membership_tables(match, seg, 10, 5, 20)
    assert membership_tables(match, seg, 10, 5, 20) is a
    match[0, 0] = -1                                  # in-place edit bumps the version -> rebuilt
    assert membership_tables(match, seg, 10, 5, 20) is not a


@pytest.mark.parametrize("C,K", [(8, 2), (64, 3), (256, 2), (60, 1)])
def test_bf16_storage_is_the_fp32_kernel_rounded_once(C, K):
    """bf16 rows in, bf16 rows out (the storage type of a bf16 model, BASELINE configs[4]): the same fp32 accumulation as
    the fp32 kernels on the same (bf16-representable) values, rounded ONCE at the store -- so the result is the fp32
    kernel's result rounded to bf16, forward and input gradient; the weight gradient (fp32 partials) is the fp32
    kernel's up to the order of its channel sum (8 channels per lane instead of 4).  C = 256 / 64 / 8 take the 16-byte
    path, C = 60 the scalar one."""
    from mlgnn.project import segment_project
    gen = torch.Generator().manual_seed(C + K)
    B, NN, G, S = 2, 40, 300, 57
    dev = "cuda:0"
    x = torch.randn(B * NN, C, generator=gen).bfloat16()
    w = torch.randn(G, K, generator=gen) * 0.3
    match = torch.randint(0, NN, (B, G), generator=gen)
    match[:, ::7] = -1
    seg = torch.sort(torch.randint(0, S, (B, G), generator=gen), dim=1)[0]
    cot = torch.randn(B, C, S, K, generator=gen).bfloat16()
    outs = []
    for dt in (torch.float32, torch.bfloat16):
        xd = x.to(dev).to(dt).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        out = segment_project(xd, match.to(dev), seg.to(dev), wd, NN, S, True)
        assert out.dtype == dt
        gx, gw = torch.autograd.grad(out, [xd, wd], cot.to(dev).to(dt))
        outs.append((out, gx, gw))
    (o32, gx32, gw32), (o16, gx16, gw16) = outs
    # (a wider row splits the members of a segment / node over the lane groups differently with 8 channels per lane than
    # with 4: the fp32 sums may differ in their last bit, which moves a bf16 rounding only at a tie)
    for got, ref in ((o16, o32), (gx16, gx32)):
        assert float((got.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
        assert float((got != ref.bfloat16()).float().mean()) < 1e-3
    assert gw16.dtype == torch.float32
    assert float((gw16 - gw32).abs().max()) <= 2e-6 * float(gw32.abs().max())


@pytest.mark.parametrize("C,K,NG", [(128, 2, 3), (32, 3, 3), (64, 1, 2)])
def test_pooled_layout_is_the_permuted_result(C, K, NG):
    """``pooled_groups = NG``: the kernels write (and read, backwards) the batch of pathway graphs ``[B * NG * K, S / NG,
    C]`` directly -- bit for bit the plain ``[B, C, S, K]`` result taken through the reference's ``reshape(B, C, S / NG,
    NG * K).permute(0, 3, 2, 1).reshape(-1, S / NG, C)`` (models/vae.py:238-243), and the same gradients."""
    from mlgnn.project import segment_project
    gen = torch.Generator().manual_seed(C + K + NG)
    dev = "cuda:0"
    B, NN, G, S = 4, 300, 2500, 146 * NG
    x = torch.randn(B * NN, C, generator=gen).to(dev)
    w = (torch.randn(G, K, generator=gen) * 0.3).to(dev)
    match = torch.randint(0, NN, (B, G), generator=gen)
    match[:, ::11] = -1
    seg = torch.sort(torch.randint(0, S, (B, G), generator=gen), dim=1)[0]
    match, seg = match.to(dev), seg.to(dev)
    cot = torch.randn(B * NG * K, S // NG, C, generator=gen).to(dev)

    def run(pooled):
        xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        out = segment_project(xd, match, seg, wd, NN, S, True, pooled_groups=NG if pooled else 0)
        if not pooled:
            out = out.reshape(B, C, S // NG, NG * K).permute(0, 3, 2, 1).reshape(-1, S // NG, C)
        gx, gw = torch.autograd.grad((out * cot).sum(), [xd, wd])
        return out.detach(), gx, gw

    o1, gx1, gw1 = run(True)
    o0, gx0, gw0 = run(False)
    assert o1.is_contiguous() and tuple(o1.shape) == (B * NG * K, S // NG, C)
    assert torch.equal(o1, o0) and torch.equal(gx1, gx0) and torch.equal(gw1, gw0)


@pytest.mark.parametrize("C,K", [(128, 2), (256, 2), (64, 3), (128, 4), (256, 3)])
def test_input_gradient_rows_with_many_and_no_members(C, K):
    """The multi-row input-gradient walks (narrow rows: 4 rows per lane group in flight; wide rows: units of eight rows whose
    membership data is fetched in one shot, eight member slots per round; 256 channels x 3: the wave-per-row form): nodes
    without a membership, nodes with one, and one node that 300 memberships point at, next to each other; the last rows of
    the table do not fill a unit."""
    from mlgnn.project import segment_project
    gen = torch.Generator().manual_seed(9)
    dev = "cuda:0"
    B, NN, G, S = 2, 1003, 3000, 438
    match = torch.randint(0, NN, (B, G), generator=gen)
    match[:, :300] = 17
    match[:, 300:900] = torch.arange(600) % 40 + 500          # rows 500..539 with 15 members each; most others 0..3
    seg = torch.sort(torch.randint(0, S, (B, G), generator=gen), dim=1)[0]
    x = torch.randn(B * NN, C, generator=gen, requires_grad=True)
    w = (torch.randn(G, K, generator=gen) * 0.3).requires_grad_(True)
    cot = torch.randn(B, C, S, K, generator=gen)
    ref = M.projection_pool(x, match, seg, w, None, NN, S, True)
    gx_ref, gw_ref = torch.autograd.grad((ref * cot).sum(), [x, w])
    xd, wd = x.detach().to(dev).requires_grad_(True), w.detach().to(dev).requires_grad_(True)
    out = segment_project(xd, match.to(dev), seg.to(dev), wd, NN, S, True)
    gx, gw = torch.autograd.grad((out * cot.to(dev)).sum(), [xd, wd])
    assert_close(gx, gx_ref, 1e-4, "grad x", elementwise=True)
    assert_close(gw, gw_ref, 1e-4, "grad w")
