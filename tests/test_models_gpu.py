"""Drop-in modules (HIP path) vs golden vectors produced by the reference's own classes, and vs
the CPU oracle.  Tolerance 1e-4 fp32 (north_star), |diff| <= 1e-4 * max(1, |ref|_inf)."""
from types import SimpleNamespace

import pytest
import torch

from _util import assert_close, golden_files, literal, load_golden, make_args

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _to_dev(ns):
    return SimpleNamespace(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in vars(ns).items()})


def _check_param_grads(module, gold, prefix="sd.", tol=TOL):
    seen = 0
    for name, p in module.named_parameters():
        key = prefix + name
        if key not in gold:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(g, gold[key], tol, "grad " + name)
        seen += 1
    assert seen > 0


@pytest.mark.parametrize("path", golden_files("genconv"))
def test_genconv_vs_reference(path):
    from models.gcn_lib.sparse.torch_vertex import GENConv
    f = load_golden(path)
    cfg = literal(f["cfg"])
    d, full = cfg.pop("d"), cfg.pop("full")
    conv = GENConv(d, d, encode_edge=True, edge_feat_dim=(d if full else 1), mlp_layers=2, **cfg)
    conv.load_state_dict(f["sd"], strict=True)
    conv.to(DEV).train()
    x = f["x"].to(DEV).requires_grad_(True)
    ea = f["edge_attr"].to(DEV).requires_grad_(True)
    out = conv(x, f["edge_index"].to(DEV), ea)
    assert_close(out, f["out"], TOL, "genconv out", elementwise=True)
    (out * f["cot"].to(DEV)).sum().backward()
    assert_close(x.grad, f["grad"]["x"], TOL, "grad x", elementwise=True)
    assert_close(ea.grad, f["grad"]["edge_attr"], TOL, "grad edge_attr")
    _check_param_grads(conv, f["grad"])


@pytest.mark.parametrize("path", golden_files("genconv"))
def test_genconv_rank_one_edge_path(path):
    """Scalar edge attribute carried factored (no [E,d] tensor): same numbers as the dense path."""
    from mlgnn import RankOneEdge
    from models.gcn_lib.sparse.torch_vertex import GENConv
    f = load_golden(path)
    cfg = literal(f["cfg"])
    d, full = cfg.pop("d"), cfg.pop("full")
    if full:
        pytest.skip("fixture with a full-width edge input")
    conv = GENConv(d, d, encode_edge=True, edge_feat_dim=1, mlp_layers=2, **cfg)
    conv.load_state_dict(f["sd"], strict=True)
    conv.to(DEV).train()
    x = f["x"].to(DEV).requires_grad_(True)
    # Linear(1 -> d) on a scalar a: a * W[:,0] + b  ==  RankOneEdge(a, 1, 0) through the encoder
    one = torch.ones(1, device=DEV)
    zero = torch.zeros(1, device=DEV)
    out = conv(x, f["edge_index"].to(DEV), RankOneEdge(f["edge_attr"][:, 0].to(DEV), one, zero))
    assert_close(out, f["out"], TOL, "genconv out (rank-1)", elementwise=True)
    (out * f["cot"].to(DEV)).sum().backward()
    assert_close(x.grad, f["grad"]["x"], TOL, "grad x", elementwise=True)
    _check_param_grads(conv, f["grad"])


@pytest.mark.parametrize("path", golden_files("sage"))
def test_sage_vs_reference(path):
    from models.gcn_lib.sparse.torch_vertex import GraphConv
    f = load_golden(path)
    cout, cin = f["sd"]["gconv.lin_r.weight"].shape
    conv = GraphConv(cin, cout, conv=str(f["kind"]), act="leakyrelu", mlp_norm="none")
    conv.load_state_dict(f["sd"], strict=True)
    conv.to(DEV)
    x = f["x"].to(DEV).requires_grad_(True)
    out = conv(x, f["edge_index"].to(DEV), f["edge_attr"].to(DEV))
    assert_close(out, f["out"], TOL, "sage out", elementwise=True)
    (out * f["cot"].to(DEV)).sum().backward()
    assert_close(x.grad, f["grad"]["x"], TOL, "grad x", elementwise=True)
    _check_param_grads(conv, f["grad"])


DEEPER_BASE = dict(num_layers=3, hidden_channels=32, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                   use_column="w", global_edge="none", graph_pooling="mean", norm="layer", mlp_layers=2,
                   block="res+", pathway_global_node=False, node_embedding=False, use_age=False,
                   num_layer_head=1, pathway_num=8, pathway_readout=None)


@pytest.mark.parametrize("path", golden_files("deepergcn"))
def test_deepergcn_vs_reference(path):
    from models import get_model
    f = load_golden(path)
    model = get_model("deepergcn")(make_args(**dict(DEEPER_BASE, **literal(f["over"]))))
    model.load_state_dict(f["sd"], strict=True)
    model.to(DEV).train()
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "batch", "age",
                                                          "pathway_node_attr", "node_size")}))
    out = model(batch)
    assert_close(out, f["out"], TOL, "deepergcn out", elementwise=True)
    (out * f["cot"].to(DEV)).sum().backward()
    _check_param_grads(model, f["grad"])


@pytest.mark.parametrize("path", golden_files("multilevel"))
def test_multilevel_vs_reference(path):
    from models import get_model
    f = load_golden(path)
    model = get_model("multilevel_gnn")(make_args(**literal(f["over"])))
    model.node_num = int(f["node_num"])
    model.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
    model.set_pca_params(torch.zeros(int((f["sd"]["info_mask"] > 0).sum()), model.pca_dim), f["sd"]["info_mask"][:, 0])
    model.set_info_mask(f["sd"]["info_mask"].clone())
    model.load_state_dict(f["sd"], strict=True)
    model.set_pathway_indexs(f["pathway_indexs"].to(DEV))
    model.to(DEV).eval()
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    pred, feat = model(batch)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature", elementwise=True)
    assert_close(pred, f["pred"], TOL, "pred", elementwise=True)
    fl = model.get_feature_loss(feat)
    assert_close(fl, f["feature_loss"], TOL, "feature loss")
    ((pred * f["cot"].to(DEV)).sum() + fl).backward()
    _check_param_grads(model, f["grad"])


@pytest.mark.parametrize("path", golden_files("mlgseq"))
def test_multilevel_seq_vs_reference(path):
    """``get_model('multilevel_gnn_seq')``: same state_dict keys as the reference (strict load), same outputs and
    parameter gradients, head in ``PathwayHeadSeq`` with and without ``only_mrna_pred``."""
    from models import get_model
    f = load_golden(path)
    model = get_model("multilevel_gnn_seq")(make_args(**literal(f["over"])))
    model.node_num = int(f["node_num"])
    model.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
    model.load_ckpt({k: torch.as_tensor(v) for k, v in f["sd"].items()})        # re-creates the projection parameter
    model.load_state_dict(f["sd"], strict=True)
    model.set_pathway_indexs(f["pathway_indexs"].to(DEV))
    model.to(DEV).eval()
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    pred, feat = model(batch)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature", elementwise=True)
    assert_close(pred, f["pred"], TOL, "pred", elementwise=True)
    fl = model.get_feature_loss(feat)
    assert_close(fl, f["feature_loss"], TOL, "feature loss")
    ((pred * f["cot"].to(DEV)).sum() + fl).backward()
    _check_param_grads(model, f["grad"])


def _vae_from_fixture(f, name="vae"):
    from models import get_model
    args = make_args(**literal(f["over"]))
    model = get_model(name)(args, None, f["pathway_indexs"])
    model.node_num = int(f["node_num"])
    model.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
    model.set_pca_params(torch.zeros(int((f["sd"]["info_mask"] > 0).sum()), model.pca_dim), f["sd"]["info_mask"][:, 0])
    model.set_info_mask(f["sd"]["info_mask"].clone())
    if name != "autoencoder":
        model.set_pathway_similarity_matrix(f["similarity"].numpy())
        model.reconstruct_head(args)
    model.load_state_dict(f["sd"], strict=True)
    model.set_pathway_indexs(f["pathway_indexs"].to(DEV))
    return model.to(DEV).eval()


@pytest.mark.parametrize("path", golden_files("vae"))
def test_vae_predict_path_vs_reference(path):
    """``VAE.train_step`` -> ``predict_head`` (DiffPool on the pathway graph at either placement, or the conv head):
    outputs, DiffPool losses and every parameter gradient against the reference's own class."""
    f = load_golden(path)
    model = _vae_from_fixture(f)
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    pred, feat, link, ent, gene = model.train_step(batch)
    assert_close(gene, f["gene_feature"], TOL, "gene_feature")
    assert_close(feat, f["pca_feature"], TOL, "pca_feature", elementwise=True)
    assert_close(pred, f["pred"], TOL, "pred", elementwise=True)
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    ((pred * f["cot"].to(DEV)).sum() + 0.7 * link + 0.3 * ent).backward()
    _check_param_grads(model, f["grad_pred"])
    ev = model.eval_step(batch)
    assert_close(ev[0], f["pred"], TOL, "eval_step pred")


@pytest.mark.parametrize("path", golden_files("vae"))
def test_vae_reconstruction_path_vs_reference(path):
    """Encoder statistics and losses, the per-pathway decoders on a given latent (batched form for the uniform
    'foreach' decoder, block loop for 'foreach_diffhidden'), KL and reconstruction terms and their gradients."""
    f = load_golden(path)
    model = _vae_from_fixture(f)
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    q_z, h, losses, _ = model.encoder(batch)
    assert_close(h, f["embedding"], TOL, "embedding")
    assert_close(losses[0], f["loss_std"], TOL, "loss_std")
    assert_close(losses[2], f["loss_corr"], TOL, "loss_corr")
    z = q_z.loc + 0.5 * q_z.scale
    assert_close(z, f["z"], TOL, "z")
    recon = model.foreach_decoder(z)
    assert_close(recon, f["recon"], TOL, "recon")
    kld = torch.distributions.kl_divergence(q_z, torch.distributions.Normal(0, 1.)).sum(-1).mean()
    rec = torch.nn.functional.mse_loss(recon, f["target"].to(DEV))
    assert_close(kld, f["kld"], TOL, "kld")
    assert_close(rec, f["rec"], TOL, "reconstruction loss")
    (rec + 0.1 * kld + losses[0] + losses[2]).backward()
    _check_param_grads(model, f["grad_rec"])
    out = model(batch)                                   # forward(): a random latent draw, shapes and finiteness
    assert out["pred_x"].shape == recon.shape and bool(torch.isfinite(out["pred_x"]).all())
    terms = model.vae_loss(out["pred_x"], f["target"].to(DEV), out["z"], out["q_z"])
    assert all(bool(torch.isfinite(v)) for v in terms.values())


@pytest.mark.parametrize("path", golden_files("vqvae"))
def test_vq_vae_vs_reference(path):
    """VQ_VAE: prediction path on the un-quantised latent, then forward() -- quantisation (same code words chosen),
    straight-through gradients, decoders, loss -- against the reference's own class."""
    f = load_golden(path)
    model = _vae_from_fixture(f, "vq_vae")
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    pred, feat, link, ent = model.train_step(batch)
    assert_close(feat, f["pca_feature"], TOL, "pca_feature", elementwise=True)
    assert_close(pred, f["pred"], TOL, "pred", elementwise=True)
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    ((pred * f["cot"].to(DEV)).sum() + 0.7 * link + 0.3 * ent).backward()
    _check_param_grads(model, f["grad_pred"])
    model.zero_grad()
    out = model(batch)
    assert_close(out["z"], f["z"], TOL, "z")
    assert_close(out["embedding"], f["quantized"], TOL, "quantized latent")
    assert_close(out["vq_loss"], f["vq_loss"], TOL, "vq_loss")
    assert_close(out["pred_x"], f["recon"], TOL, "recon")
    terms = model.vae_loss(out["pred_x"], f["target"].to(DEV), out["vq_loss"])
    assert_close(terms["loss"], f["loss"], TOL, "loss")
    terms["loss"].backward()
    _check_param_grads(model, f["grad_rec"])


@pytest.mark.parametrize("path", golden_files("autoencoder"))
def test_autoencoder_vs_reference(path):
    f = load_golden(path)
    model = _vae_from_fixture(f, "autoencoder")
    batch = _to_dev(SimpleNamespace(**{k: f[k] for k in ("x", "edge_index", "edge_attr", "gene_pca_match",
                                                          "raw_indice", "age")}))
    recon, h, none = model(batch)
    assert none is None
    assert_close(h, f["latent"], TOL, "latent")
    assert_close(recon, f["recon"], TOL, "recon")
    (recon * f["cot"].to(DEV)).sum().backward()
    _check_param_grads(model, f["grad"])


@pytest.mark.parametrize("path", golden_files("diffpool"))
def test_diffpool_vs_reference(path):
    from models import DiffPool
    f = load_golden(path)
    Bp, C, hid, outc, nl, apl = [int(v) for v in f["cfg"]]
    dp = DiffPool(C, None, 146, nl, hid, outc, SimpleNamespace(pooling_type="correlation", after_pooling_layer=apl))
    dp.load_state_dict({k: torch.as_tensor(v) for k, v in f["sd"].items()}, strict=True)
    dp.to(DEV).eval()
    x = f["x"].to(DEV).requires_grad_(True)
    out, link, ent = dp(x, f["adj"].to(DEV))
    assert_close(out, f["out"], TOL, "diffpool out")
    assert_close(link, f["link"], TOL, "link")
    assert_close(ent, f["ent"], TOL, "ent")
    ((out * f["cot"].to(DEV)).sum() + 0.7 * link + 0.3 * ent).backward()
    assert_close(x.grad, f["grad"]["x"], TOL, "grad x", elementwise=True)
    _check_param_grads(dp, f["grad"])
