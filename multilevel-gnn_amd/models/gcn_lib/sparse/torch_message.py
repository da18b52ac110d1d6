"""Aggregator front-end (interface of the reference's ``models/gcn_lib/sparse/torch_message.py``:
``GenMessagePassing`` :8-85, ``MsgNorm`` :168-179).

The reference derives from PyG ``MessagePassing`` and reduces a materialised ``[E, d]`` message
tensor with torch_scatter.  Here the class only carries the aggregator configuration and its
learnable scalars; ``reduce_messages`` hands the whole ``message + aggregate`` step to ONE fused
HIP kernel (``mlgnn_csr_aggregate_fwd``) over the CSR graph.
"""
import torch
import torch.nn.functional as F
from torch import nn

from mlgnn import gen_aggregate

_SOFTMAX = ("softmax_sg", "softmax", "softmax_sum")
_POWER = ("power", "power_sum")
_PLAIN = ("add", "mean", "max")


class GenMessagePassing(nn.Module):
    def __init__(self, aggr='softmax', t=1.0, learn_t=False, p=1.0, learn_p=False, y=0.0, learn_y=False):
        super().__init__()
        if aggr not in _SOFTMAX + _POWER + _PLAIN:
            raise NotImplementedError('To be implemented')
        self.aggr = aggr
        self.learn_t = False
        self.learn_p = False
        if aggr in _SOFTMAX:
            if learn_t and aggr in ('softmax', 'softmax_sum'):
                self.learn_t = True
                self.t = nn.Parameter(torch.Tensor([t]), requires_grad=True)
            else:
                self.t = t
        elif aggr in _POWER:
            if learn_p:
                self.learn_p = True
                self.p = nn.Parameter(torch.Tensor([p]), requires_grad=True)
            else:
                self.p = p
        if aggr in ('softmax_sum', 'power_sum'):
            self.y = nn.Parameter(torch.Tensor([y]), requires_grad=learn_y)

    def reduce_messages(self, x, graph, edge, eps, add_root=False):
        """``aggregate(relu(x_j + e_ij) + eps)`` for every destination node (torch_message.py:44-85);
        ``add_root`` returns ``x + aggregate`` (GENConv's ``h``) from the same kernel pass."""
        scaled = self.aggr in ('softmax_sum', 'power_sum')
        out = gen_aggregate(x, graph, edge, aggr=self.aggr, t=getattr(self, "t", 1.0), p=getattr(self, "p", 1.0),
                            eps=eps, learn_t=self.learn_t, learn_p=self.learn_p, add_root=add_root and not scaled)
        if scaled:
            self.sigmoid_y = torch.sigmoid(self.y)
            out = torch.pow(graph.in_degree.unsqueeze(1), self.sigmoid_y) * out
            if add_root:
                out = out + x
        return out


class MsgNorm(nn.Module):
    def __init__(self, learn_msg_scale=False):
        super().__init__()
        self.msg_scale = nn.Parameter(torch.Tensor([1.0]), requires_grad=learn_msg_scale)

    def forward(self, x, msg, p=2):
        msg = F.normalize(msg, p=p, dim=1)
        return msg * x.norm(p=p, dim=1, keepdim=True) * self.msg_scale
