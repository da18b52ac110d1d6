"""Dense building blocks of the graph convolutions (interface of the reference's
``models/gcn_lib/sparse/torch_nn.py``: ``act_layer`` :9-24, ``norm_layer`` :27-38, ``MLP`` :54-75).

These are the dense epilogues of every conv (GEMM + norm + activation); they run on
rocBLAS/hipBLASLt and MIOpen through torch.  The OGB ``AtomEncoder``/``BondEncoder`` tables of
the reference are molecule-dataset leftovers outside the hot path and are not provided.
"""
from torch import nn

from mlgnn.dense import fused_mlp2, fused_mlp2_post_supported, fused_mlp2_supported, linear
from mlgnn.norm import layer_norm_act, layer_norm_act_fork

_ACTS = {
    "relu": lambda inplace, slope, n: nn.ReLU(inplace),
    "leakyrelu": lambda inplace, slope, n: nn.LeakyReLU(slope, inplace),
    "prelu": lambda inplace, slope, n: nn.PReLU(num_parameters=n, init=slope),
    "elu": lambda inplace, slope, n: nn.ELU(),
    "tanh": lambda inplace, slope, n: nn.Tanh(),
}

_NORMS = {
    "batch": lambda nc: nn.BatchNorm1d(nc, affine=True),
    "layer": lambda nc: nn.LayerNorm(nc, elementwise_affine=True),
    "instance": lambda nc: nn.InstanceNorm1d(nc, affine=False),
}


def act_layer(act_type, inplace=False, neg_slope=0.2, n_prelu=1):
    try:
        return _ACTS[act_type.lower()](inplace, neg_slope, n_prelu)
    except KeyError:
        raise NotImplementedError("activation layer [%s] is not found" % act_type)


def norm_layer(norm_type, nc):
    try:
        return _NORMS[norm_type.lower()](nc)
    except KeyError:
        raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def _enabled(name):
    return name is not None and isinstance(name, str) and name.lower() != "none"


def _post(out, post_norm):
    """``(identity, relu?(norm(out)))`` as a separate (fork) pass: shapes the fused epilogue does not cover."""
    pn, relu = post_norm
    y, identity = layer_norm_act_fork(out, pn.weight, pn.bias, pn.eps, relu=relu)
    return identity, y


class MLP(nn.Sequential):
    """``Linear -> [norm] -> [act] -> [Dropout2d]`` per hop; the last hop is a bare Linear when
    ``last_lin``.  Child indices (and so ``state_dict`` keys) follow the reference's layout."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0., last_lin=False):
        layers = []
        hops = len(channels) - 1
        for h in range(hops):
            layers.append(nn.Linear(channels[h], channels[h + 1], bias))
            if last_lin and h == hops - 1:
                break
            if _enabled(norm):
                layers.append(norm_layer(norm, channels[h + 1]))
            if _enabled(act):
                layers.append(act_layer(act))
            if drop > 0:
                layers.append(nn.Dropout2d(drop))
        super().__init__(*layers)

    def forward(self, x, residual=None, post_norm=None):
        """Same children, same order; a ``LayerNorm`` directly followed by ``ReLU`` runs as ONE
        fused HIP pass (``mlgnn.norm.layer_norm_act``) instead of two ATen passes; ``nn.Linear``
        children use ``mlgnn.dense.linear`` (split-precision MFMA GEMMs); ``residual`` is added in the last
        Linear's epilogue.  ``post_norm = (nn.LayerNorm, relu)``: the caller's next step is that norm (+ ReLU) of the
        result -- returns ``(out, relu?(norm(out)))``, computed in the last GEMM's epilogue when the fused MLP applies."""
        mods = list(self)
        # Linear -> LayerNorm -> ReLU -> Linear (GENConv's MLP): one fused op, no LayerNorm pass in between
        if (len(mods) == 4 and type(mods[0]) is nn.Linear and isinstance(mods[1], nn.LayerNorm)
                and mods[1].elementwise_affine and isinstance(mods[2], nn.ReLU) and type(mods[3]) is nn.Linear
                and fused_mlp2_supported(x, mods[0].weight, mods[3].weight)):
            if post_norm is not None and fused_mlp2_post_supported(x, mods[0].weight, mods[3].weight, post_norm[0].weight):
                pn, relu = post_norm
                return fused_mlp2(x, mods[0].weight, mods[0].bias, mods[1].weight, mods[1].bias, mods[1].eps,
                                  mods[3].weight, mods[3].bias, residual, (pn.weight, pn.bias, pn.eps, relu))
            out = fused_mlp2(x, mods[0].weight, mods[0].bias, mods[1].weight, mods[1].bias, mods[1].eps,
                             mods[3].weight, mods[3].bias, residual)
            return out if post_norm is None else _post(out, post_norm)
        if post_norm is not None:
            return _post(self.forward(x, residual), post_norm)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.LayerNorm) and m.elementwise_affine and x.dim() == 2:
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = layer_norm_act(x, m.weight, m.bias, m.eps, relu)
                i += 2 if relu else 1
            elif type(m) is nn.Linear:
                last = i == len(mods) - 1
                x = linear(x, m.weight, m.bias, residual if last else None)
                if last:
                    residual = None
                i += 1
            else:
                x = m(x)
                i += 1
        return x if residual is None else x + residual
