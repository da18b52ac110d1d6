"""Shared helpers for the test-suite (fixture loading, namespaces)."""
import ast
import glob
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")),
                  key=lambda p: int(os.path.basename(p)[len(prefix) + 1:-4]))


def load_golden(path):
    """-> (top-level dict of tensors/scalars, nested dicts for 'sd/', 'grad/')."""
    if not os.path.isabs(path):
        path = os.path.join(GOLDEN, path)
    z = np.load(path, allow_pickle=False)
    top, nested = {}, {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "fiub":
            val = torch.from_numpy(v.copy()) if v.ndim > 0 else torch.tensor(v.item())
        else:
            val = v.item() if v.ndim == 0 else v
        if "/" in k:
            head, rest = k.split("/", 1)
            nested.setdefault(head, {})[rest] = val
        else:
            top[k] = val
    for head, d in nested.items():                       # dicts stored packed (make_golden.py::save)
        if "__names__" in d:
            flat, off = d["__flat__"], 0
            un = {k: v for k, v in d.items() if not k.startswith("__")}
            for name, shape in zip(d["__names__"].tolist(), d["__shapes__"].tolist()):
                shape = [n for n in shape if n >= 0]
                n = int(np.prod(shape)) if shape else 1
                un[str(name)] = flat[off:off + n].reshape(shape).clone()
                off += n
            assert off == flat.numel()
            nested[head] = un
    top.update(nested)
    return top


def literal(s):
    return dict(ast.literal_eval(str(s)))


# Defaults of the reference's opt.py for the flags the hot-path models read
# (opt.py:88-198,350-354,415-428); tests override per fixture.
OPT_DEFAULTS = dict(
    num_layers=3, mlp_layers=2, hidden_channels=128, block="res+", conv="gen", gcn_aggr="max", norm="layer",
    num_tasks=2, t=1.0, p=1.0, learn_t=False, learn_p=False, msg_norm=False, learn_msg_scale=False,
    conv_encode_edge=False, graph_pooling="mean", node_embedding=False, node_num=5606, node_embedding_dim=32,
    num_layer_head=1, use_age=False, head_dropout=False, use_edge_attr=False, pathway_readout="maxpool",
    gnn_encoder="linear", pca_only=False, no_inter_drop=False, no_inter_norm=False, head_init=False,
    all_init=True, pre_readout_drop=False, pre_concat_age=False, global_edge="onehot", init_emb=False,
    feature_drop=False, dropout=0.5, mul_attr=False, pathway_global_node=False, pathway_num=146,
    use_column=None, pathway_edge_num=8,
    # MultilevelGNN
    resgnn=False, pca_match_mask=False, final_channels=1, final_head=1, used_omics="012", only_mrna_pred=False, vqvae_num_embeddings=512, channel_one=False, vae_generate_train_sample=False, decoder_dim=4096, decoder_type='flatten', pathway_similarity='correlation', std_weight=False, grad_weight=False, mmd_kernel_type='imq', mmd_alpha=-9.0, mmd_beta=10.5, kld_weight=0.2, mmd_reg_weight=110, z_var=2, std_weight_coef=1, grad_weight_coef=1, load_autoencoder_epoch=None, autoencoder_ckpt_path=None,
    pca_compare=False,
    pca_prelinear=False, learnable_pca=False, pca_loss=False, pca_loss_coef=1.0, pca_indep_loss=False,
    pca_init_type=None, pca_dim=2, pca_pool_dim=2, mutual_info_mask=False, mutual_info_threshold=None,
    pathway_pool_dim=4, freeze_pca_weight=False, value_att_mask=False, node_select_threshold=1,
    mutual_neighbors=3, freeze_node_embedding=False, head_dim=64, gnn_name="gat", dense_gnn=False,
    weighted_edge=False, gnn_act="leakyrelu", reorder_pathway=False, reorder_type="pca", gnn_last_norm=False,
    gnn_mlp_norm="none", merge_mode="mult", add_coef1=0.5, add_coef2=0.5, repeat_mask=False, repeat_cyclic=2,
    repeat_norm=False, conv_channel_list=[32, 64], conv_kernel_list=[1, 1], embedding_init_type="xavier",
    emb_val=0.01, input_drop=None, input_emb_drop=None, gnn_dropout=0.0, device_num=1, edge_type="grnboost2",
    reduction_method="linear_projection", diff_pooling_location="pathway", diff_pooling_layer=2,
    diff_pooling_hidden_dim=32, diff_pooling_output_dim=64, after_pooling_layer=1, pooling_type="correlation",
    freeze_mutual_select_init=False, random_state=12345, remain_all_tf=False, device=0,
)


def make_args(**over):
    d = dict(OPT_DEFAULTS)
    d.update(over)
    return SimpleNamespace(**d)


def assert_close(a, b, tol=1e-4, what="", elementwise=False):
    """Default (parameter gradients, scalars): the norm form |a-b| <= tol * max(1, |b|_inf).
    ``elementwise=True`` (forward outputs, gradients w.r.t. the inputs): every entry on its own,
    |a-b| <= tol * |b| + 0.1 * tol * max(1, |b|_inf) -- relative ``tol`` per element with an absolute floor of a
    tenth of the norm form's allowance, so an entry of size 1e-3 next to one of size 300 is still held to ~3e-3
    absolute instead of 3e-2."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    assert bool(torch.isfinite(a).all()) == bool(torch.isfinite(b).all()), "%s: non-finite entries differ" % what
    scale = max(1.0, float(b.abs().max()))
    if elementwise:
        bound = tol * b.abs() + 0.1 * tol * scale
        over = (a - b).abs() - bound
        worst = int(over.argmax())
        assert float(over.max()) <= 0.0, "%s: entry %d: |%.6e - %.6e| > %.3e (elementwise, tol %.1e)" % (
            what, worst, float(a.flatten()[worst]), float(b.flatten()[worst]), float(bound.flatten()[worst]), tol)
        return
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max|diff|=%.3e > %.1e*%.3g" % (what, err, tol, scale)
