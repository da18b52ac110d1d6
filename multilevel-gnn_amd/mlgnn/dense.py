"""Dense pooled-graph operators of DiffPool (reference: PyG 2.2.0 ``DenseSAGEConv`` and
``dense_diff_pool`` as called from ``models/diff_pooling.py:24-32,36,45,64``)."""
import torch
import torch.nn.functional as F

DIFFPOOL_EPS = 1e-15


def dense_sage(x, adj, w_rel, w_root, b_root, normalize=True):
    """``normalize(W_rel (A x / clamp(rowsum A, 1)) + W_root x + b)``; 2-D ``adj`` broadcasts."""
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    agg = torch.matmul(adj, x) / adj.sum(dim=-1, keepdim=True).clamp(min=1)
    out = F.linear(agg, w_rel) + F.linear(x, w_root, b_root)
    return F.normalize(out, p=2.0, dim=-1) if normalize else out


def dense_diff_pool(z, adj, s):
    """``S = softmax(s)``; returns ``(S^T Z, S^T A S, ||A - S S^T||_F / numel(A), mean entropy)``."""
    z = z.unsqueeze(0) if z.dim() == 2 else z
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    s = s.unsqueeze(0) if s.dim() == 2 else s
    s = torch.softmax(s, dim=-1)
    st = s.transpose(1, 2)
    out = torch.matmul(st, z)
    out_adj = torch.matmul(torch.matmul(st, adj), s)
    link = torch.norm(adj - torch.matmul(s, st), p=2) / adj.numel()
    ent = (-s * torch.log(s + DIFFPOOL_EPS)).sum(dim=-1).mean()
    return out, out_adj, link, ent
