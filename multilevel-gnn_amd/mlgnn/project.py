"""Gene -> pathway learnable-projection pooling (reference ``models/multilevel_gnn.py:212-239``).

``out[b, c, s, k] = sum_{g : raw_indice[b,g] = s} x[b, match[b,g], c] * [match >= 0] * P[g,k] * mask[g]``

The reference materialises ``[B, G, C, k]`` three times (gather, repeat, permute) and reduces it
with an atomic ``scatter_reduce``.  Here the gather is reduced per projection column without the
``k``-fold blow-up.
"""
import torch


def segment_project(x_nodes, gene_pca_match, raw_indice, weights, nodes_per_graph, n_segments,
                    match_mask=True):
    """``x_nodes [B*NN, C]`` -> ``[B, C, n_segments, k]``; ``weights [G, k]`` already carries the
    info mask.  A negative ``match`` wraps exactly as the reference's advanced indexing does when
    ``match_mask`` is off."""
    B, G = gene_pca_match.shape
    C = x_nodes.shape[1]
    k = weights.shape[1]
    total = x_nodes.shape[0]
    offs = torch.arange(B, device=x_nodes.device)[:, None] * nodes_per_graph
    idx = torch.remainder(gene_pca_match + offs, total).reshape(-1)
    xg = x_nodes.index_select(0, idx).reshape(B, G, C)
    if match_mask:
        xg = xg * (gene_pca_match >= 0).to(xg.dtype)[:, :, None]
    seg = (raw_indice.to(torch.long) + torch.arange(B, device=x_nodes.device)[:, None] * n_segments).reshape(-1)
    cols = []
    for j in range(k):
        contrib = (xg * weights[:, j][None, :, None]).reshape(B * G, C)
        cols.append(x_nodes.new_zeros((B * n_segments, C)).index_add_(0, seg, contrib))
    out = torch.stack(cols, dim=-1).reshape(B, n_segments, C, k)
    return out.permute(0, 2, 1, 3)
