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
