"""Reference graph-conv library, restated as pure functions of a ``state_dict``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Every function takes the
parameters as a flat ``{key: tensor}`` mapping with the reference's own
``state_dict`` key names under ``prefix`` and executes the reference's
*literal* op sequence -- materialised ``[E, d]`` gather, elementwise message,
scatter reduce, per-edge GEMM for SAGE -- so that timing it is timing the
reference-semantics CPU path.

Conventions: ``edge_index[0] = src = j``, ``edge_index[1] = dst = i`` (PyG
``flow='source_to_target'``); messages are reduced over ``dst``.
"""
import torch
import torch.nn.functional as F

from . import primitives as P


# ----------------------------------------------------------------------------
# models/gcn_lib/sparse/torch_nn.py
# ----------------------------------------------------------------------------
def act(x, name, neg_slope=0.2):
    """``act_layer`` (torch_nn.py:9-24); prelu needs a parameter and is not on the path."""
    name = name.lower()
    if name == "relu":
        return F.relu(x)
    if name == "leakyrelu":
        return F.leaky_relu(x, neg_slope)
    if name == "elu":
        return F.elu(x)
    if name == "tanh":
        return torch.tanh(x)
    raise NotImplementedError(name)


def norm(x, kind, sd, prefix, training=False):
    """``norm_layer`` (torch_nn.py:27-38) applied to a 2-D ``[N, C]`` input."""
    kind = kind.lower()
    if kind == "layer":
        return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)
    if kind == "batch":
        rm, rv = sd.get(prefix + "running_mean"), sd.get(prefix + "running_var")
        # functional batch norm without touching the caller's running stats
        return F.batch_norm(x, None if training else rm, None if training else rv,
                            sd[prefix + "weight"], sd[prefix + "bias"], training or rm is None, 0.1, 1e-5)
    raise NotImplementedError(kind)


def mlp(x, sd, prefix, channels, act_name="relu", norm_kind=None, last_lin=False, training=False):
    """``MLP`` (torch_nn.py:54-75): ``Linear -> [norm] -> [act]`` per hop, last hop bare if
    ``last_lin``.  Sequential indices advance exactly as the reference's list ``m`` does
    (dropout p=0 on the path, so no Dropout2d entries)."""
    has_norm = norm_kind is not None and isinstance(norm_kind, str) and norm_kind.lower() != "none"
    has_act = act_name is not None and act_name.lower() != "none"
    k = 0
    for i in range(1, len(channels)):
        x = F.linear(x, sd[prefix + "%d.weight" % k], sd.get(prefix + "%d.bias" % k))
        k += 1
        if i == len(channels) - 1 and last_lin:
            continue
        if has_norm:
            x = norm(x, norm_kind, sd, prefix + "%d." % k, training)
            k += 1
        if has_act:
            x = act(x, act_name)
            k += 1
    return x


# ----------------------------------------------------------------------------
# models/gcn_lib/sparse/torch_message.py
# ----------------------------------------------------------------------------
def gen_aggregate(msg, index, dim_size, aggr, t=1.0, learn_t=False, p=1.0, y=None):
    """``GenMessagePassing.aggregate`` (torch_message.py:44-85).

    ``t``/``p``/``y`` are floats or 1-element tensors (parameters when learnt).
    ``softmax``/``softmax_sg``: weights computed under ``no_grad`` unless
    ``learn_t`` (:51-55).  ``power``: the reference clamps the message
    IN PLACE (:70) -- zero gradient outside ``[1e-7, 10]``.
    """
    if aggr in ("add", "sum"):
        return P.scatter_sum(msg, index, dim_size)
    if aggr == "mean":
        return P.scatter_mean(msg, index, dim_size)
    if aggr == "max":
        return P.scatter_max(msg, index, dim_size)[0]
    if aggr in ("softmax", "softmax_sg", "softmax_sum"):
        if learn_t:
            w = P.scatter_softmax(msg * t, index, dim_size)
        else:
            with torch.no_grad():
                w = P.scatter_softmax(msg * t, index, dim_size)
        out = P.scatter_sum(msg * w, index, dim_size)
        if aggr == "softmax_sum":
            deg = P.degree(index, dim_size, msg.dtype).unsqueeze(1)
            out = torch.pow(deg, torch.sigmoid(y)) * out
        return out
    if aggr in ("power", "power_sum"):
        lo, hi = 1e-7, 1e1
        msg = msg.clamp(lo, hi)
        out = P.scatter_mean(torch.pow(msg, p), index, dim_size)
        out = out.clamp(lo, hi)
        out = torch.pow(out, 1 / p)
        if aggr == "power_sum":
            deg = P.degree(index, dim_size, msg.dtype).unsqueeze(1)
            out = torch.pow(deg, torch.sigmoid(y)) * out
        return out
    raise NotImplementedError(aggr)


def msg_norm(x, msg, scale, p=2):
    """``MsgNorm.forward`` (torch_message.py:175-179)."""
    msg = F.normalize(msg, p=p, dim=1)
    return msg * x.norm(p=p, dim=1, keepdim=True) * scale


# ----------------------------------------------------------------------------
# models/gcn_lib/sparse/torch_vertex.py
# ----------------------------------------------------------------------------
def genconv(x, edge_index, edge_attr, sd, prefix, *, aggr="softmax", t=1.0, learn_t=False,
            p=1.0, learn_p=False, msg_norm_on=False, encode_edge=False, norm_kind="batch",
            mlp_layers=2, eps=1e-7, training=False):
    """``GENConv.forward`` + ``message`` (torch_vertex.py:72-101), ``gnn_encoder='linear'``.

    ``e = edge_encoder(edge_attr)`` if ``encode_edge`` (:76-77);
    ``msg = relu(x_j + e) + eps`` (:94-101); ``m = aggregate(msg)``;
    ``m = MsgNorm(x, m)`` if enabled (:86-87); ``out = MLP(x + m)`` (:89-90).
    """
    N, d = x.shape
    if encode_edge and edge_attr is not None:
        e = F.linear(edge_attr, sd[prefix + "edge_encoder.weight"], sd[prefix + "edge_encoder.bias"])
    else:
        e = edge_attr
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    z = x_j + e.flatten(1) if e is not None else x_j
    msg = F.relu(z) + eps
    tt = sd[prefix + "t"] if learn_t and aggr in ("softmax", "softmax_sum") else t
    pp = sd[prefix + "p"] if learn_p and aggr in ("power", "power_sum") else p
    yy = sd.get(prefix + "y")
    m = gen_aggregate(msg, dst, N, aggr, t=tt,
                      learn_t=learn_t and aggr in ("softmax", "softmax_sum"), p=pp, y=yy)
    if msg_norm_on:
        m = msg_norm(x, m, sd[prefix + "msg_norm.msg_scale"])
    h = x + m
    channels = [d] + [2 * d] * (mlp_layers - 1) + [sd[prefix + "feature_encoder.%d.weight"
                                                      % _last_lin_index(mlp_layers, norm_kind)].shape[0]]
    return mlp(h, sd, prefix + "feature_encoder.", channels, "relu", norm_kind, last_lin=True,
               training=training)


def _last_lin_index(mlp_layers, norm_kind):
    has_norm = norm_kind is not None and str(norm_kind).lower() != "none"
    per_hop = 1 + (1 if has_norm else 0) + 1
    return per_hop * (mlp_layers - 1)


def sageconv(x, edge_index, edge_attr, sd, prefix, *, act_name="leakyrelu", relative=False,
             normalize=False, mlp_norm=None, training=False):
    """``SAGEConv.forward/message/update`` via ``RSAGEConv`` (torch_vertex.py:269-304).

    Drop self loops, append self loops with attribute 1.0 (:272-273); per-edge
    GEMM ``(x_j * w) @ lin_r.weight.T`` (``- x_i`` first if ``relative``)
    (:279-286); PyG ``SAGEConv`` default ``aggr='mean'`` over ``dst``;
    ``update``: ``nn(cat(x, aggr))`` with ``nn = MLP([in+out, out], act)``
    (:288-291, :302); ``lin_l`` and ``bias`` are unused (``bias=False``).
    """
    N = x.shape[0]
    ei, ea = P.remove_self_loops(edge_index, edge_attr)
    ei, ea = P.add_self_loops(ei, ea, 1.0, N)
    if ea is not None and ea.dim() == 1:
        ea = ea.unsqueeze(-1)
    src, dst = ei[0], ei[1]
    x_j = x.index_select(0, src)
    if ea is not None:
        x_j = x_j * ea
    w_r = sd[prefix + "lin_r.weight"]
    if relative:
        msg = torch.matmul(x_j - x.index_select(0, dst), w_r.t())
    else:
        msg = torch.matmul(x_j, w_r.t())
    aggr_out = P.scatter_mean(msg, dst, N)
    out_c = w_r.shape[0]
    out = mlp(torch.cat((x, aggr_out), dim=1), sd, prefix + "nn.", [x.shape[1] + out_c, out_c],
              act_name, mlp_norm, training=training)
    if normalize:
        out = F.normalize(out, p=2.0, dim=-1)
    return out
