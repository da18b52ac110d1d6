"""The oracle restatement vs golden vectors produced by the reference's own classes
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle."""
from types import SimpleNamespace

import pytest
import torch

from oracle import gcn_lib as G
from oracle import models as M
from _util import assert_close, golden_files, literal, load_golden, make_args

TOL = 2e-5      # oracle and reference run the same torch CPU ops; only op order differs


def _leaf(t):
    if not torch.is_tensor(t):
        return torch.as_tensor(t)
    return t.clone().requires_grad_(True) if t.dtype.is_floating_point else t


def _sd_leaves(sd):
    """Parameters become leaves; buffers (BatchNorm running stats, info_mask) stay plain."""
    return {k: (v if ("running_" in k or k == "info_mask" or "num_batches" in k) else _leaf(v))
            for k, v in sd.items()}


def _check_grads(loss, named, gold, tol=TOL):
    names = [k for k in gold if k in named and named[k].requires_grad]
    assert names
    gs = torch.autograd.grad(loss, [named[k] for k in names], allow_unused=True)
    for k, g in zip(names, gs):
        g = torch.zeros_like(named[k]) if g is None else g
        assert_close(g, gold[k], tol, "grad " + k)


def test_aggregators_match_reference():
    f = load_golden("aggregators.npz")
    N = int(f["n_nodes"])
    for ci in range(int(f["n_cases"])):
        c = f["c%d" % ci]
        aggr, kw = str(c["aggr"]), literal(c["kw"])
        inp = _leaf(f["inputs"])
        named = {"inputs": inp}
        t = kw.get("t", 1.0)
        if kw.get("learn_t") and aggr in ("softmax", "softmax_sum"):
            t = named["t"] = _leaf(torch.tensor([t]))
        p = kw.get("p", 1.0)
        if kw.get("learn_p"):
            p = named["p"] = _leaf(torch.tensor([p]))
        y = None
        if aggr.endswith("_sum"):
            y = named["y"] = _leaf(torch.tensor([kw.get("y", 0.0)]))
        out = G.gen_aggregate(inp * 1.0, f["index"], N, aggr, t=t, learn_t=bool(kw.get("learn_t")), p=p, y=y)
        assert_close(out, c["out"], TOL, "aggr %s" % aggr)
        gold = {k[len("grad/"):]: v for k, v in c.items() if k.startswith("grad/")}
        _check_grads((out * c["cot"]).sum(), named, gold)


@pytest.mark.parametrize("path", golden_files("genconv"))
def test_genconv_matches_reference(path):
    f = load_golden(path)
    cfg = literal(f["cfg"])
    sd = _sd_leaves(f["sd"])
    x, ea = _leaf(f["x"]), _leaf(f["edge_attr"])
    out = G.genconv(x, f["edge_index"], ea, sd, "", aggr=cfg["aggr"], t=cfg.get("t", 1.0),
                    learn_t=cfg.get("learn_t", False), p=cfg.get("p", 1.0), learn_p=cfg.get("learn_p", False),
                    msg_norm_on=cfg.get("msg_norm", False),
                    encode_edge=True, norm_kind=cfg["norm"], mlp_layers=2, training=True)
    assert_close(out, f["out"], TOL, "genconv out")
    named = {"x": x, "edge_attr": ea}
    named.update({"sd." + k: v for k, v in sd.items()})
    _check_grads((out * f["cot"]).sum(), named, f["grad"])


@pytest.mark.parametrize("path", golden_files("sage"))
def test_sage_matches_reference(path):
    f = load_golden(path)
    sd = _sd_leaves(f["sd"])
    x = _leaf(f["x"])
    out = G.sageconv(x, f["edge_index"], f["edge_attr"], sd, "gconv.", relative=(str(f["kind"]) == "rsage"))
    assert_close(out, f["out"], TOL, "sage out")
    named = {"x": x}
    named.update({"sd." + k: v for k, v in sd.items()})
    _check_grads((out * f["cot"]).sum(), named, f["grad"])
    assert float(f["grad"]["sd.gconv.lin_l.weight"].abs().max()) == 0.0      # dead parameter


def _batch(f):
    return SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "batch", "age", "pathway_node_attr",
                                                "node_size", "gene_pca_match", "raw_indice") if k in f})


DEEPER_BASE = dict(num_layers=3, hidden_channels=32, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                   use_column="w", global_edge="none", graph_pooling="mean", norm="layer", mlp_layers=2,
                   block="res+", pathway_global_node=False, node_embedding=False, use_age=False,
                   num_layer_head=1, pathway_num=8, pathway_readout=None)


@pytest.mark.parametrize("path", golden_files("deepergcn"))
def test_deepergcn_matches_reference(path):
    f = load_golden(path)
    args = make_args(**dict(DEEPER_BASE, **literal(f["over"])))
    sd = _sd_leaves(f["sd"])
    out = M.deepergcn_forward(args, sd, _batch(f), training=True)
    assert_close(out, f["out"], TOL, "deepergcn out")
    _check_grads((out * f["cot"]).sum(), {"sd." + k: v for k, v in sd.items()}, f["grad"])


@pytest.mark.parametrize("path", golden_files("multilevel"))
def test_multilevel_matches_reference(path):
    f = load_golden(path)
    args = make_args(**literal(f["over"]))
    sd = _sd_leaves(f["sd"])
    pred, feat = M.multilevel_gnn_forward(args, sd, _batch(f), int(f["node_num"]), training=False)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature")
    assert_close(pred, f["pred"], TOL, "pred")
    fl = M.feature_loss(args, sd, feat, f["pathway_indexs"])
    assert_close(fl, f["feature_loss"], TOL, "feature loss")
    _check_grads((pred * f["cot"]).sum() + fl, {"sd." + k: v for k, v in sd.items()}, f["grad"])


@pytest.mark.parametrize("path", golden_files("mlgseq"))
def test_multilevel_seq_matches_reference(path):
    """MultilevelGNNSeq (PathwayHeadSeq head, with and without only_mrna_pred) vs the reference's own classes."""
    f = load_golden(path)
    args = make_args(**literal(f["over"]))
    sd = _sd_leaves(f["sd"])
    assert any(k.startswith("pathwayhead.") for k in sd) and not any(k.startswith("conv_model.") for k in sd)
    pred, feat = M.multilevel_gnn_seq_forward(args, sd, _batch(f), int(f["node_num"]), training=False)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature")
    assert_close(pred, f["pred"], TOL, "pred")
    fl = M.feature_loss(args, sd, feat, f["pathway_indexs"])
    assert_close(fl, f["feature_loss"], TOL, "feature loss")
    _check_grads((pred * f["cot"]).sum() + fl, {"sd." + k: v for k, v in sd.items()}, f["grad"])


@pytest.mark.parametrize("path", golden_files("vae"))
def test_vae_matches_reference(path):
    """VAE (vae.py): encoder statistics and losses, per-pathway decoders on a given latent, train_step ->
    predict_head with DiffPool at either placement or the conv / max-pool head; values and parameter gradients."""
    f = load_golden(path)
    args = make_args(**literal(f["over"]))
    sd = _sd_leaves(f["sd"])
    named = {"sd." + k: v for k, v in sd.items()}
    adj = (f["similarity"] + torch.eye(146, dtype=f["similarity"].dtype)).to(torch.float32)
    pred, feat, link, ent, gene = M.vae_train_step(args, sd, _batch(f), int(f["node_num"]), adj)
    assert_close(gene, f["gene_feature"], TOL, "gene_feature")
    assert_close(feat, f["pca_feature"], TOL, "pca_feature")
    assert_close(pred, f["pred"], TOL, "pred")
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    _check_grads((pred * f["cot"]).sum() + 0.7 * link + 0.3 * ent, named, f["grad_pred"])
    mu, sigma, losses, _ = M.vae_encoder(args, sd, _batch(f), int(f["node_num"]))
    assert_close(torch.cat([mu, sigma], -1), f["embedding"], TOL, "embedding")
    assert_close(losses[0], f["loss_std"], TOL, "loss_std")
    assert_close(losses[2], f["loss_corr"], TOL, "loss_corr")
    recon = M.vae_foreach_decoder(sd, mu + 0.5 * (sigma + 1e-7))
    assert_close(mu + 0.5 * (sigma + 1e-7), f["z"], TOL, "z")
    assert_close(recon, f["recon"], TOL, "recon")
    q_z = torch.distributions.Normal(mu, sigma + 1e-7)
    kld = torch.distributions.kl_divergence(q_z, torch.distributions.Normal(0, 1.)).sum(-1).mean()
    rec = torch.nn.functional.mse_loss(recon, f["target"])
    assert_close(kld, f["kld"], TOL, "kld")
    assert_close(rec, f["rec"], TOL, "reconstruction loss")
    _check_grads(rec + 0.1 * kld + losses[0] + losses[2], named, f["grad_rec"])


@pytest.mark.parametrize("path", golden_files("vqvae"))
def test_vq_vae_matches_reference(path):
    """VQ_VAE (vq_vae.py): train_step -> predict_head on the un-quantised latent; forward = encoder, nearest-code
    quantisation with the straight-through estimator, decoders; loss = mmd_beta * mse + vq_loss."""
    f = load_golden(path)
    args = make_args(**literal(f["over"]))
    sd = _sd_leaves(f["sd"])
    named = {"sd." + k: v for k, v in sd.items()}
    adj = (f["similarity"] + torch.eye(146)).to(torch.float32)
    pred, feat, link, ent = M.vq_vae_train_step(args, sd, _batch(f), int(f["node_num"]), adj)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature")
    assert_close(pred, f["pred"], TOL, "pred")
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    _check_grads((pred * f["cot"]).sum() + 0.7 * link + 0.3 * ent, named, f["grad_pred"])
    recon, qz, z, vq_loss = M.vq_vae_forward(args, sd, _batch(f), int(f["node_num"]))
    assert_close(z, f["z"], TOL, "z")
    assert_close(qz, f["quantized"], TOL, "quantized")
    assert len(torch.unique(qz.reshape(-1, qz.shape[-1]), dim=0)) > 1          # more than one code word in use
    assert_close(vq_loss, f["vq_loss"], TOL, "vq_loss")
    assert_close(recon, f["recon"], TOL, "recon")
    loss = args.mmd_beta * torch.nn.functional.mse_loss(recon, f["target"]) + vq_loss
    assert_close(loss, f["loss"], TOL, "loss")
    _check_grads(loss, named, f["grad_rec"])


@pytest.mark.parametrize("path", golden_files("autoencoder"))
def test_autoencoder_matches_reference(path):
    f = load_golden(path)
    args = make_args(**literal(f["over"]))
    sd = _sd_leaves(f["sd"])
    recon, h = M.autoencoder_forward(args, sd, _batch(f), int(f["node_num"]), args.decoder_type)
    assert_close(h, f["latent"], TOL, "latent")
    assert_close(recon, f["recon"], TOL, "recon")
    _check_grads((recon * f["cot"]).sum(), {"sd." + k: v for k, v in sd.items()}, f["grad"])


@pytest.mark.parametrize("path", golden_files("diffpool"))
def test_diffpool_matches_reference(path):
    f = load_golden(path)
    Bp, C, hid, outc, nl, apl = [int(v) for v in f["cfg"]]
    sd = _sd_leaves(f["sd"])
    x = _leaf(f["x"])
    out, link, ent = M.diffpool_forward(sd, x, f["adj"], nl, apl)
    assert_close(out, f["out"], TOL, "diffpool out")
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    named = {"x": x}
    named.update({"sd." + k: v for k, v in sd.items()})
    _check_grads((out * f["cot"]).sum() + 0.7 * link + 0.3 * ent, named, f["grad"])
    assert M.diffpool_cluster_sizes(146, 2) == [37, 10]
