"""Sparse graph convolutions on the HIP CSR kernels (interface of the reference's
``models/gcn_lib/sparse/torch_vertex.py``: ``GENConv`` :12-104, ``SAGEConv`` :226-294,
``RSAGEConv`` :297-304, ``GraphConv`` :338-363).

Only the conv types reachable from the shipped configs are provided (``gen``, ``sage``, ``rsage``);
the PyG-wrapper types (edge/mr/gat/gcn/gin) raise ``NotImplementedError``.
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from mlgnn import LowRankEdge, TableEdge, as_graph, weighted_mean_aggregate
from mlgnn.dense import linear
from mlgnn.graph import sage_graph
from mlgnn.norm import msg_norm_add
from .torch_message import GenMessagePassing, MsgNorm
from .torch_nn import MLP

_SAGE_FUSED = os.environ.get("MLGNN_SAGE_FUSED", "1") == "1"      # (0: the separate aggregate / lin_r / cat / Linear / act ops, for A/B runs)


class GENConv(GenMessagePassing):
    """GENeralized graph convolution: ``MLP(x + [MsgNorm] aggregate(relu(x_j + e_ij) + eps))``."""

    def __init__(self, in_dim, emb_dim, aggr='softmax', t=1.0, learn_t=False, p=1.0, learn_p=False,
                 y=0.0, learn_y=False, gnn_encoder='linear', msg_norm=False, learn_msg_scale=True,
                 encode_edge=False, bond_encoder=False, edge_feat_dim=None, norm='batch', mlp_layers=2,
                 eps=1e-7, pca_only=False):
        super().__init__(aggr=aggr, t=t, learn_t=learn_t, p=p, learn_p=learn_p, y=y, learn_y=learn_y)
        if gnn_encoder != 'linear':
            raise NotImplementedError("gnn_encoder=%r is outside the accelerated path" % (gnn_encoder,))
        if bond_encoder:
            raise NotImplementedError("OGB bond encoder is outside the accelerated path")
        self.gnn_encoder = gnn_encoder
        self.feature_encoder = MLP([in_dim] + [in_dim * 2] * (mlp_layers - 1) + [emb_dim], norm=norm, last_lin=True)
        self.eps = eps
        self.encode_edge = encode_edge
        self.bond_encoder = bond_encoder
        self.msg_norm = MsgNorm(learn_msg_scale=learn_msg_scale) if msg_norm else None
        if encode_edge:
            self.edge_encoder = nn.Linear(edge_feat_dim, in_dim)
        self.pca_only = pca_only

    def forward(self, x, edge_index, edge_attr=None, residual=None, post_norm=None):
        """``edge_index``: COO ``[2, E]`` or a prebuilt :class:`mlgnn.CSRGraph`.
        ``residual``: added to the result inside the last Linear's epilogue (the caller's ``conv(...) + h``).
        ``post_norm = (nn.LayerNorm, relu)``: the caller's next step is ``relu?(norm(result))`` (the res+ block's
        pre-conv norm, deepergcn.py:236-241) -- returns ``(result, relu?(norm(result)))``, the second value written by
        the last Linear's epilogue instead of by a pass of its own.
        ``edge_attr``: ``[E, d_e]`` tensor, a :class:`mlgnn.LowRankEdge` (raw attributes kept factored
        through the Linear encoders: no ``[E, d]`` tensor, no edge GEMM) or a :class:`mlgnn.TableEdge`
        (one row of a small table per edge: the edge-type embedding)."""
        if self.pca_only:
            return self.feature_encoder(x)
        graph = as_graph(edge_index, x.shape[0])
        if isinstance(edge_attr, (LowRankEdge, TableEdge)):
            edge = edge_attr.through_linear(self.edge_encoder.weight, self.edge_encoder.bias) \
                if self.encode_edge else edge_attr
        elif edge_attr is not None:
            edge = self.edge_encoder(edge_attr) if self.encode_edge else edge_attr
            if edge.dim() != 2:                   # (a 2-D embedding is passed on as the object it is: it may carry the
                edge = edge.flatten(1)            #  shared-gradient tag of mlgnn.share_edge_gradient)
        else:
            edge = None
        flat = x.flatten(1)
        if self.msg_norm is None:
            h = self.reduce_messages(flat, graph, edge, self.eps, add_root=True)       # x + m in one pass
        else:
            h = msg_norm_add(flat, self.reduce_messages(flat, graph, edge, self.eps), self.msg_norm.msg_scale)
        if h.shape != x.shape:                    # (a same-shape reshape would drop the row-max tag of the kernel's output)
            h = h.reshape(x.shape)
        if isinstance(self.feature_encoder, MLP) and (residual is not None or post_norm is not None):
            return self.feature_encoder(h, residual=residual, post_norm=post_norm)
        out = self.feature_encoder(h)
        return out if residual is None else out + residual


class Linear(nn.Module):
    """Bias-optional affine map that is deliberately NOT an ``nn.Linear`` subclass, like the PyG
    ``Linear`` the reference's SAGE base class creates: the models' xavier ``init_weight`` sweeps
    (multilevel_gnn.py:294-299) skip it, so it keeps kaiming-uniform(a=sqrt 5) initial values."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class SAGEConv(nn.Module):
    """Weighted-mean GraphSAGE: ``nn(cat(x, mean_j((x_j * w_ij [- x_i]) W_r^T)))``.

    The reference multiplies every EDGE by ``W_r`` before the mean (torch_vertex.py:279-286);
    the mean is linear, so the kernel reduces ``x_j * w_ij`` first and ``W_r`` is applied to ``N``
    rows instead of ``E + N``.  ``lin_l`` exists only for ``state_dict`` compatibility (the
    reference never uses it either)."""

    def __init__(self, in_channels, out_channels, nn, norm=True, bias=True, relative=False, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.relative = relative
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)
        self.nn = nn
        self.normalize = norm
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.bias = None

    def _fused_update(self):
        """``(slope,)`` when ``self.nn`` is the shipped ``Linear -> LeakyReLU / ReLU`` (or a bare Linear) and nothing else
        of this conv stands in the way of the one-product form (mlgnn.sage); None otherwise."""
        mods = list(self.nn) if isinstance(self.nn, nn.Sequential) else None
        if mods is None or self.normalize or self.bias is not None or not mods or type(mods[0]) is not nn.Linear:
            return None
        if len(mods) == 1:
            return (1.0,)
        if len(mods) == 2 and isinstance(mods[1], nn.LeakyReLU) and mods[1].negative_slope > 0:
            return (float(mods[1].negative_slope),)
        if len(mods) == 2 and type(mods[1]) is nn.ReLU:
            return (0.0,)
        return None

    def forward(self, x, edge_index, size=None, edge_attr=None, row_scale=None, shared=None):
        """``row_scale`` [N] or None: the caller's next step is ``out * row_scale[:, None]`` (MultilevelGNN's value mask,
        multilevel_gnn.py:205-207) -- applied here, in the update's epilogue when the fused layer runs.
        ``shared``: the batch's :class:`mlgnn.graph.SharedTopology` (every sample carries the same graph) or None."""
        if size is not None:
            raise NotImplementedError("bipartite propagation is outside the accelerated path")
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        graph, weight = sage_graph(edge_index, edge_attr, x.shape[0], shared)
        fused = self._fused_update() if _SAGE_FUSED else None
        if fused is not None and (row_scale is None or not row_scale.requires_grad):
            from mlgnn.sage import sage_layer, sage_layer_supported
            lin = self.nn[0]
            if sage_layer_supported(x, lin.weight, self.lin_r.weight, False):
                return sage_layer(x, graph, weight, lin.weight, lin.bias, self.lin_r.weight, fused[0], self.relative,
                                  row_scale)
        out = self._forward_unfused(x, graph, weight)
        return out if row_scale is None else out * row_scale.reshape(-1, 1)

    def _forward_unfused(self, x, graph, weight):
        agg = weighted_mean_aggregate(x, graph, weight, mean=True)
        if self.relative:
            agg = agg - x                    # every node has its self loop: mean_j(x_i) = x_i
        aggr_out = linear(agg, self.lin_r.weight)          # tall-matrix weight gradient on the MFMA kernel
        if self.bias is not None:
            aggr_out = aggr_out + self.bias
        out = self.nn(torch.cat((x, aggr_out), dim=1))
        if self.normalize:
            out = F.normalize(out, p=2, dim=-1)
        return out


class RSAGEConv(SAGEConv):
    def __init__(self, in_channels, out_channels, act='relu', norm=False, mlp_norm=None, bias=True,
                 relative=False, drop=0.0):
        nn_ = MLP([out_channels + in_channels, out_channels], act, mlp_norm, bias, drop=drop)
        super().__init__(in_channels, out_channels, nn_, norm, False, relative)


class GraphConv(nn.Module):
    """Static graph convolution dispatcher (torch_vertex.py:338-363)."""

    def __init__(self, in_channels, out_channels, conv='edge', act='relu', norm=None, bias=True, heads=8,
                 mlp_norm=None, drop=0.0):
        super().__init__()
        kind = conv.lower()
        if kind == 'sage':
            self.gconv = RSAGEConv(in_channels, out_channels, act, norm, mlp_norm, bias, False, drop)
        elif kind == 'rsage':
            self.gconv = RSAGEConv(in_channels, out_channels, act, norm, mlp_norm, bias, True, drop)
        elif kind in ('edge', 'mr', 'gat', 'gcn', 'gin'):
            raise NotImplementedError('conv {} needs PyG layers outside the accelerated path'.format(conv))
        else:
            raise NotImplementedError('conv {} is not implemented'.format(conv))

    def forward(self, x, edge_index, edge_attr=None, row_scale=None, shared=None):
        return self.gconv(x, edge_index, edge_attr=edge_attr, row_scale=row_scale, shared=shared)
