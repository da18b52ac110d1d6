"""Per-graph readout over the ``batch`` vector (reference: PyG ``global_{add,mean,max}_pool``
called from ``deepergcn.py:148-155,319``; torch_scatter semantics: an empty graph yields 0)."""
import torch


def global_pool(x, batch, kind, num_graphs=None):
    """``x [N, d]``, ``batch [N]`` graph id per node -> ``[B, d]``.  ``num_graphs`` avoids the
    device->host sync of ``batch.max() + 1`` the reference pays."""
    B = int(num_graphs) if num_graphs is not None else int(batch.max().item()) + 1
    batch = batch.to(torch.long)
    if kind in ("sum", "add", "mean"):
        out = x.new_zeros((B, x.shape[1])).index_add_(0, batch, x)
        if kind == "mean":
            cnt = x.new_zeros(B).index_add_(0, batch, x.new_ones(x.shape[0]))
            out = out / cnt.clamp(min=1)[:, None]
        return out
    if kind == "max":
        idx = batch[:, None].expand_as(x)
        out = x.new_full((B, x.shape[1]), float("-inf")).scatter_reduce(0, idx, x, reduce="amax", include_self=True)
        return torch.where(torch.isinf(out), torch.zeros_like(out), out)
    raise ValueError(kind)
