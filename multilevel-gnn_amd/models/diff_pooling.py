"""DiffPool on MI355X (interface of the reference's ``models/diff_pooling.py``:
``SAGEConvolutions`` :11-46, ``DiffPoolLayer`` :49-65, ``DiffPool`` :68-133).

The dense GraphSAGE layers and the soft-assignment contraction (``S^T Z``, ``S^T A S``, link and
entropy losses -- PyG ``DenseSAGEConv`` / ``dense_diff_pool`` in the reference) go through
:mod:`mlgnn.dense`.  Same constructors, ``forward(x, adj) -> (x, link_total, ent_total)`` and
``state_dict`` keys (``...lin_rel.weight``, ``...lin_root.{weight,bias}``, ``bns.*``).
"""
from math import ceil

import torch
from torch import nn
from torch.nn import functional as F

from mlgnn.dense import dense_diff_pool, dense_sage
from .gcn_lib.sparse.torch_vertex import Linear


class DenseSAGEConv(nn.Module):
    """``normalize(lin_rel(A x / clamp(rowsum A, 1)) + lin_root(x))`` -- PyG 2.2.0 DenseSAGEConv."""

    def __init__(self, in_channels, out_channels, normalize=False, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.normalize = in_channels, out_channels, normalize
        self.lin_rel = Linear(in_channels, out_channels, bias=False)
        self.lin_root = Linear(in_channels, out_channels, bias=bias)

    def forward(self, x, adj, mask=None):
        if mask is not None:
            raise NotImplementedError("node masks are not used on the reference's call path (vae.py:238-243)")
        return dense_sage(x, adj, self.lin_rel.weight, self.lin_root.weight, self.lin_root.bias, self.normalize)


class SAGEConvolutions(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, residual=True):
        super().__init__()
        self.num_layers = num_layers
        self.residual = residual
        self.layers = nn.ModuleList()
        self.bns = nn.ModuleList()
        width = in_channels
        for _ in range(num_layers - 1):
            self.layers.append(DenseSAGEConv(width, out_channels, normalize=True))
            self.bns.append(nn.BatchNorm1d(out_channels))
            width = out_channels
        self.layers.append(DenseSAGEConv(width, out_channels, normalize=True))

    def forward(self, x, adj, mask=None):
        for i in range(self.num_layers - 1):
            x_new = F.relu(self.layers[i](x, adj, mask))
            b, n, c = x_new.size()
            x_new = self.bns[i](x_new.view(-1, c)).view(b, n, c)
            x = x + x_new if (self.residual and x.shape == x_new.shape) else x_new
        return self.layers[self.num_layers - 1](x, adj, mask)


class DiffPoolLayer(nn.Module):
    def __init__(self, dim_input, dim_embedding, current_num_clusters, no_new_clusters):
        super().__init__()
        self.gnn_pool = SAGEConvolutions(1, dim_input, no_new_clusters)
        self.gnn_embed = SAGEConvolutions(1, dim_input, dim_embedding)

    def forward(self, x, adj, mask=None):
        s = self.gnn_pool(x, adj, mask)
        z = self.gnn_embed(x, adj, mask)
        return dense_diff_pool(z, adj, s)


class DiffPool(nn.Module):
    def __init__(self, num_features, num_classes, max_num_nodes, num_layers, gnn_hidden_dim, gnn_output_dim,
                 args, encode_edge=False, pre_sum_aggr=False):
        super().__init__()
        if pre_sum_aggr:
            raise NotImplementedError("pre_sum_aggr (DenseGraphConv, IMDB only) is outside the accelerated path")
        self.args = args
        self.encode_edge = encode_edge
        self.max_num_nodes = max_num_nodes
        self.pooling_type = args.pooling_type
        self.num_pooling_layers = num_layers
        coarse = 0.1 if num_layers == 1 else 0.25
        # constructed-but-never-called in the reference too; kept for state_dict compatibility
        self.initial_embed = SAGEConvolutions(1, num_features, gnn_output_dim)
        layers, after = [], []
        clusters, new_clusters = max_num_nodes, ceil(coarse * max_num_nodes)
        for i in range(num_layers):
            cin = num_features if i == 0 else gnn_hidden_dim
            cout = gnn_output_dim if i == num_layers - 1 else gnn_hidden_dim
            layers.append(DiffPoolLayer(cin, cout, clusters, new_clusters))
            clusters, new_clusters = new_clusters, ceil(new_clusters * coarse)
            after.append(SAGEConvolutions(args.after_pooling_layer, cout, cout))
        self.diffpool_layers = nn.ModuleList(layers)
        self.after_pool_layers = nn.ModuleList(after)

    def forward(self, x, adj, mask=None):
        l_total, e_total = 0, 0
        for i in range(self.num_pooling_layers):
            x, adj, l, e = self.diffpool_layers[i](x, adj, mask if i == 0 else None)
            x = self.after_pool_layers[i](x, adj)
            l_total = l_total + l
            e_total = e_total + e
        return x, l_total, e_total
