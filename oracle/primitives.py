"""Third-party primitives of the reference's hot path, restated in plain torch.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The reference calls these
through ``torch_scatter==2.1.0`` and ``torch_geometric==2.2.0``
(``requirements.txt:149-150``), neither of which is vendored or installable
here, so each function follows the published algorithm of that pinned version
and names the reference call site that relies on it.  **parity unpinned** for
this file: the reference tree contains no test or golden vector at this
boundary.

All functions are differentiable with the same sub-gradient choices as the
pinned wheels' CPU paths (``scatter_max`` routes the gradient to the FIRST
maximal element of each group).
"""
import torch


def _expand_index(index, src):
    # index [E] -> broadcast over trailing dims of src [E, ...]
    view = [-1] + [1] * (src.dim() - 1)
    return index.view(view).expand_as(src)


def scatter_sum(src, index, dim_size):
    """``torch_scatter.scatter(src, index, dim=0, dim_size, reduce='sum')``.

    Call sites: ``models/gcn_lib/sparse/torch_message.py:57`` and, through
    PyG's ``SumAggregation``, ``torch_message.py:47``.
    """
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.index_add(0, index, src)


def degree(index, num_nodes, dtype=torch.float32):
    """``torch_geometric.utils.degree`` (``torch_message.py:62,79``): float count."""
    out = torch.zeros(num_nodes, dtype=dtype, device=index.device)
    return out.index_add(0, index, torch.ones(index.numel(), dtype=dtype, device=index.device))


def scatter_mean(src, index, dim_size):
    """``scatter(..., reduce='mean')``: sum / clamp(count, 1)  (``torch_message.py:71``,
    PyG ``MeanAggregation`` used by ``tg.nn.SAGEConv`` -- ``torch_vertex.py:226``)."""
    total = scatter_sum(src, index, dim_size)
    count = degree(index, dim_size, src.dtype).clamp(min=1)
    return total / count.view([-1] + [1] * (src.dim() - 1))


def scatter_max(src, index, dim_size):
    """``torch_scatter.scatter_max`` CPU semantics: returns ``(out, arg)``.

    * groups with no element give ``out = 0`` and ``arg = E`` (the wheel fills
      ``arg`` with ``src.size(dim)`` and masks ``out`` to 0 there);
    * the first element (lowest edge position) attaining the maximum wins
      (the CPU loop updates only on a strict ``>``);
    * the gradient flows to that single element;
    * a NaN element never wins: the CPU loop starts from ``numeric_limits::lowest()`` and updates on
      ``new > current``, which is false for NaN (a group of NaNs only keeps its fill value and comes out as 0).

    Call sites: PyG ``MaxAggregation`` from ``torch_message.py:47``;
    ``global_max_pool`` (``deepergcn.py:153``); inside ``scatter_softmax``.
    """
    E = src.shape[0]
    idx = _expand_index(index, src)
    with torch.no_grad():
        init = src.new_full((dim_size,) + tuple(src.shape[1:]), float("-inf"))
        comparable = torch.where(torch.isnan(src), torch.full_like(src, float("-inf")), src)
        vmax = init.scatter_reduce(0, idx, comparable, reduce="amax", include_self=True)
        pos = torch.arange(E, device=src.device).view([-1] + [1] * (src.dim() - 1)).expand_as(src)
        is_max = src == vmax.gather(0, idx)
        cand = torch.where(is_max, pos, torch.full_like(pos, E))
        arg = torch.full((dim_size,) + tuple(src.shape[1:]), E, dtype=torch.long, device=src.device)
        arg = arg.scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
    empty = arg == E
    safe = arg.clamp(max=max(E - 1, 0))
    if E == 0:
        return src.new_zeros((dim_size,) + tuple(src.shape[1:])), arg
    out = src.gather(0, safe)
    out = torch.where(empty, torch.zeros_like(out), out)
    return out, arg


def scatter_min(src, index, dim_size):
    out, arg = scatter_max(-src, index, dim_size)
    return -out, arg


def scatter(src, index, dim_size, reduce="sum"):
    """Dispatcher with the wheel's reduce names (``sum``/``add``/``mean``/``max``/``min``)."""
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim_size)
    if reduce == "max":
        return scatter_max(src, index, dim_size)[0]
    if reduce == "min":
        return scatter_min(src, index, dim_size)[0]
    raise ValueError("unknown reduce %r" % (reduce,))


def scatter_softmax(src, index, dim_size):
    """``torch_scatter.composite.scatter_softmax`` as of 2.1.0: no epsilon.

    ``exp(src - max_group) / sum_group exp(src - max_group)``, per trailing
    channel.  Call sites: ``torch_message.py:52,55``.
    """
    gmax, _ = scatter_max(src, index, dim_size)
    centred = src - gmax.index_select(0, index)
    ex = centred.exp()
    denom = scatter_sum(ex, index, dim_size).index_select(0, index)
    return ex / denom


def remove_self_loops(edge_index, edge_attr=None):
    """``torch_geometric.utils.remove_self_loops`` (``torch_vertex.py:272``)."""
    keep = edge_index[0] != edge_index[1]
    ei = edge_index[:, keep]
    return (ei, None) if edge_attr is None else (ei, edge_attr[keep])


def add_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    """``torch_geometric.utils.add_self_loops`` (``torch_vertex.py:273``): appends
    ``(i, i)`` for every node AFTER the existing edges; attributes filled with 1.0."""
    N = int(num_nodes)
    loop = torch.arange(N, dtype=edge_index.dtype, device=edge_index.device)
    ei = torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)
    if edge_attr is None:
        return ei, None
    fill = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fill_value)
    return ei, torch.cat([edge_attr, fill], dim=0)


def global_pool(x, batch, kind, size=None):
    """``global_{add,mean,max}_pool`` (``deepergcn.py:148-155,319``): scatter over ``batch``
    with ``dim_size = batch.max() + 1``."""
    B = int(batch.max().item()) + 1 if size is None else int(size)
    red = {"sum": "sum", "add": "sum", "mean": "mean", "max": "max"}[kind]
    return scatter(x, batch, B, red)


def dense_sage_conv(x, adj, w_rel, w_root, b_root, normalize=True):
    """``torch_geometric.nn.DenseSAGEConv.forward`` (2.2.0), mask=None.

    ``out = lin_rel(adj @ x / clamp(adj.sum(-1, keepdim), min=1)) + lin_root(x)``;
    ``lin_rel`` has no bias, ``lin_root`` carries the bias; L2-normalise the
    channel dimension when ``normalize``.  A 2-D ``adj`` is unsqueezed and
    broadcast over the batch.  Call sites: ``diff_pooling.py:24-32,36,45``.
    """
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    out = torch.matmul(adj, x)
    out = out / adj.sum(dim=-1, keepdim=True).clamp(min=1)
    out = torch.nn.functional.linear(out, w_rel) + torch.nn.functional.linear(x, w_root, b_root)
    if normalize:
        out = torch.nn.functional.normalize(out, p=2.0, dim=-1)
    return out


DIFFPOOL_EPS = 1e-15


def dense_diff_pool(x, adj, s, normalize=True):
    """``torch_geometric.nn.dense_diff_pool`` (2.2.0), mask=None (``diff_pooling.py:64``).

    ``S = softmax(s, -1)``; ``X' = S^T X``; ``A' = S^T A S``;
    ``link = ||A - S S^T||_F`` (``/ adj.numel()`` when ``normalize``);
    ``ent = mean_over_nodes(sum_k -S log(S + 1e-15))``.
    """
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    s = s.unsqueeze(0) if s.dim() == 2 else s
    s = torch.softmax(s, dim=-1)
    st = s.transpose(1, 2)
    out = torch.matmul(st, x)
    out_adj = torch.matmul(torch.matmul(st, adj), s)
    link = adj - torch.matmul(s, st)
    link = torch.norm(link, p=2)
    if normalize:
        link = link / adj.numel()
    ent = (-s * torch.log(s + DIFFPOOL_EPS)).sum(dim=-1).mean()
    return out, out_adj, link, ent
