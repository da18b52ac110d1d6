"""Oracle primitives vs hand-computed cases (the third-party boundary is 'parity unpinned':
these closed-form checks are all that pins it).  CPU only."""
import math

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
