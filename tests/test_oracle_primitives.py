"""Oracle primitives vs hand-computed cases (the third-party boundary is 'parity unpinned':
these closed-form checks are all that pins it).  CPU only."""
import math

import pytest

import torch

from oracle import primitives as P
from _util import assert_close


def test_scatter_sum_mean_empty_groups():
    src = torch.tensor([[1., 2.], [3., 4.], [5., 6.]])
    idx = torch.tensor([2, 0, 2])
    assert_close(P.scatter_sum(src, idx, 4), [[3, 4], [0, 0], [6, 8], [0, 0]])
    assert_close(P.scatter_mean(src, idx, 4), [[3, 4], [0, 0], [3, 4], [0, 0]])


def test_scatter_max_first_wins_and_empty_is_zero():
    src = torch.tensor([[-1.0], [5.0], [5.0], [-3.0]], requires_grad=True)
    idx = torch.tensor([1, 0, 0, 1])
    out, arg = P.scatter_max(src, idx, 3)
    assert out.tolist() == [[5.0], [-1.0], [0.0]]
    assert arg.tolist() == [[1], [0], [4]]              # tie -> first edge; empty -> E
    out.sum().backward()
    assert src.grad.tolist() == [[1.0], [1.0], [0.0], [0.0]]


def test_scatter_softmax_closed_form_no_eps():
    src = torch.tensor([[0.0], [math.log(3.0)], [7.0]])
    idx = torch.tensor([0, 0, 1])
    w = P.scatter_softmax(src, idx, 2)
    assert_close(w, [[0.25], [0.75], [1.0]], 1e-6)


def test_self_loops():
    ei = torch.tensor([[0, 1, 2, 2], [1, 1, 0, 2]])
    ea = torch.tensor([[.1], [.2], [.3], [.4]])
    ei2, ea2 = P.remove_self_loops(ei, ea)
    assert ei2.tolist() == [[0, 2], [1, 0]]
    assert_close(ea2.flatten(), [.1, .3], 1e-7)
    ei3, ea3 = P.add_self_loops(ei2, ea2, 1.0, 3)
    assert ei3.tolist() == [[0, 2, 0, 1, 2], [1, 0, 0, 1, 2]]
    assert_close(ea3.flatten(), [.1, .3, 1., 1., 1.], 1e-7)


def test_degree_and_pools():
    b = torch.tensor([0, 0, 1])
    x = torch.tensor([[1., -2.], [3., -4.], [5., 6.]])
    assert P.degree(b, 3).tolist() == [2.0, 1.0, 0.0]
    assert_close(P.global_pool(x, b, "mean"), [[2, -3], [5, 6]])
    assert_close(P.global_pool(x, b, "max"), [[3, -2], [5, 6]])
    assert_close(P.global_pool(x, b, "sum"), [[4, -6], [5, 6]])


def test_dense_sage_conv_formula():
    x = torch.tensor([[[1., 0.], [0., 2.]]])
    adj = torch.tensor([[0., 4.], [0., 0.]])          # row sums 4 and 0 -> clamp(0,1)=1
    w_rel = torch.eye(2)
    w_root = 2 * torch.eye(2)
    b = torch.tensor([1., 1.])
    out = P.dense_sage_conv(x, adj, w_rel, w_root, b, normalize=False)
    # node0: (4*[0,2])/4 + 2*[1,0] + 1 = [3,3]; node1: 0/1 + [0,4] + 1 = [1,5]
    assert_close(out, [[[3., 3.], [1., 5.]]], 1e-6)
    outn = P.dense_sage_conv(x, adj, w_rel, w_root, b, normalize=True)
    assert_close(outn.norm(dim=-1), [[1., 1.]], 1e-6)


def test_dense_diff_pool_formula():
    x = torch.tensor([[[1., 2.], [3., 4.]]])
    adj = torch.tensor([[1., 1.], [1., 1.]])
    s = torch.zeros(1, 2, 2)                           # softmax -> 0.5 everywhere
    out, oadj, link, ent = P.dense_diff_pool(x, adj, s)
    assert_close(out, [[[2., 3.], [2., 3.]]], 1e-6)
    assert_close(oadj, [[[1., 1.], [1., 1.]]], 1e-6)
    # A - S S^T = 1 - 0.5 = 0.5 each; ||.||_F = 1.0; /numel(adj unsqueezed)=4
    assert_close(link, 0.25, 1e-6)
    assert_close(ent, math.log(2.0), 1e-6)


def test_scatter_max_ignores_nan_like_the_comparison_loop():
    """torch_scatter's CPU scatter_max updates on ``new > current`` from ``lowest()``: NaN never wins."""
    nan = float("nan")
    src = torch.tensor([[1.0, nan], [nan, nan], [3.0, 2.0], [nan, 5.0]])
    index = torch.tensor([0, 0, 0, 1])
    out, arg = P.scatter_max(src, index, 3)
    assert torch.equal(out, torch.tensor([[3.0, 2.0], [0.0, 5.0], [0.0, 0.0]]))     # all-NaN group -> fill -> 0
    assert torch.equal(arg, torch.tensor([[2, 2], [4, 3], [4, 4]]))


# ---- second, independent formulation of every third-party primitive (fp64, dense masks / explicit loops) ----------
# The restatements in oracle/primitives.py are built on index_add / scatter_reduce.  The reference holds no vector at
# this boundary ("parity unpinned"), so each one is additionally held to a formulation that shares no code with it:
# a dense [N, E] membership mask, or a Python loop over groups.  This does not pin the semantics to the wheels; it
# removes the single-implementation risk.

def _rand_groups(seed, N=23, E=160, d=5):
    g = torch.Generator().manual_seed(seed)
    index = torch.randint(0, N - 3, (E,), generator=g)             # the last three groups stay empty
    src = torch.randn(E, d, generator=g, dtype=torch.float64)
    src[7] = src[3]                                                # an exact tie inside group index[3] (if shared)
    index[7] = index[3]
    mask = torch.zeros(N, E, dtype=torch.float64)
    mask[index, torch.arange(E)] = 1.0
    return src, index, mask, N


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scatter_family_against_dense_masks(seed):
    src, index, mask, N = _rand_groups(seed)
    cnt = mask.sum(1, keepdim=True)
    assert torch.allclose(P.scatter_sum(src, index, N), mask @ src, atol=1e-12)
    assert torch.allclose(P.scatter_mean(src, index, N), (mask @ src) / cnt.clamp(min=1), atol=1e-12)
    assert torch.equal(P.degree(index, N, torch.float64), cnt.squeeze(1))
    big = torch.where(mask[:, :, None] > 0, src[None], torch.full((), -float("inf"), dtype=torch.float64))   # [N,E,d]
    vmax, amax = big.max(dim=1)                                    # torch.max returns the first maximal position
    want = torch.where(cnt > 0, vmax, torch.zeros_like(vmax))
    out, arg = P.scatter_max(src, index, N)
    assert torch.equal(out, want)
    assert torch.equal(arg[cnt.squeeze(1) > 0], amax[cnt.squeeze(1) > 0])
    assert bool((arg[cnt.squeeze(1) == 0] == src.shape[0]).all())
    # softmax: per group and channel, exp(x - max) / sum, by explicit loops
    sm = P.scatter_softmax(src, index, N)
    for n in range(N):
        rows = (index == n).nonzero().squeeze(1)
        if rows.numel():
            assert torch.allclose(sm[rows], torch.softmax(src[rows], dim=0), atol=1e-12)
    assert torch.allclose(mask @ sm, (cnt > 0).to(torch.float64).expand(-1, src.shape[1]), atol=1e-12)


def test_scatter_max_gradient_goes_to_the_first_maximal_element():
    src = torch.tensor([[1.0], [4.0], [4.0], [2.0]], dtype=torch.float64, requires_grad=True)
    out, _ = P.scatter_max(src, torch.tensor([0, 0, 0, 1]), 2)
    out.sum().backward()
    assert torch.equal(src.grad, torch.tensor([[0.0], [1.0], [0.0], [1.0]], dtype=torch.float64))


def test_self_loop_helpers_and_global_pool_against_loops():
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, 9, (2, 40), generator=g)
    ea = torch.randn(40, 2, generator=g)
    kept = [(int(s), int(t), ea[i]) for i, (s, t) in enumerate(ei.t().tolist()) if s != t]
    e2, a2 = P.remove_self_loops(ei, ea)
    assert e2.t().tolist() == [[s, t] for s, t, _ in kept] and torch.equal(a2, torch.stack([a for _, _, a in kept]))
    e3, a3 = P.add_self_loops(e2, a2, 1.0, 9)
    assert e3[:, -9:].tolist() == [list(range(9)), list(range(9))] and bool((a3[-9:] == 1.0).all())
    x = torch.randn(30, 4, generator=g, dtype=torch.float64)
    batch = torch.sort(torch.randint(0, 5, (30,), generator=g))[0]
    B = int(batch.max()) + 1
    for kind, fn in (("sum", lambda t: t.sum(0)), ("mean", lambda t: t.mean(0)), ("max", lambda t: t.max(0)[0])):
        want = torch.stack([fn(x[batch == b]) if bool((batch == b).any()) else torch.zeros(4, dtype=torch.float64)
                            for b in range(B)])
        assert torch.allclose(P.global_pool(x, batch, kind), want, atol=1e-12)


def test_dense_sage_and_diff_pool_against_explicit_loops():
    g = torch.Generator().manual_seed(6)
    B, n, C, O, K = 2, 7, 3, 4, 3
    x = torch.randn(B, n, C, generator=g, dtype=torch.float64)
    adj = torch.rand(n, n, generator=g, dtype=torch.float64) * (torch.rand(n, n, generator=g) > 0.4)
    wr, wo = torch.randn(O, C, generator=g, dtype=torch.float64), torch.randn(O, C, generator=g, dtype=torch.float64)
    b = torch.randn(O, generator=g, dtype=torch.float64)
    want = torch.zeros(B, n, O, dtype=torch.float64)
    for bb in range(B):
        for i in range(n):
            agg = sum(adj[i, j] * x[bb, j] for j in range(n)) / max(float(adj[i].sum()), 1.0)
            o = wr @ agg + wo @ x[bb, i] + b
            want[bb, i] = o / max(float(o.norm()), 1e-12)
    assert torch.allclose(P.dense_sage_conv(x, adj, wr, wo, b, normalize=True), want, atol=1e-12)
    s = torch.randn(B, n, K, generator=g, dtype=torch.float64)
    S = torch.softmax(s, -1)
    out, oadj, link, ent = P.dense_diff_pool(x, adj, s)
    sq, e_sum = 0.0, 0.0
    for bb in range(B):
        for k in range(K):
            assert torch.allclose(out[bb, k], sum(S[bb, i, k] * x[bb, i] for i in range(n)), atol=1e-12)
            for l in range(K):
                val = sum(S[bb, i, k] * adj[i, j] * S[bb, j, l] for i in range(n) for j in range(n))
                assert abs(float(oadj[bb, k, l]) - float(val)) < 1e-12
        for i in range(n):
            for j in range(n):
                sq += float(adj[i, j] - (S[bb, i] * S[bb, j]).sum()) ** 2
            e_sum += float(-(S[bb, i] * torch.log(S[bb, i] + 1e-15)).sum())
    assert abs(float(link) - sq ** 0.5 / adj.numel()) < 1e-12        # numel of the [1,n,n] adjacency as passed
    assert abs(float(ent) - e_sum / (B * n)) < 1e-12
