"""BASELINE configs[2] counterpart: MultilevelGNN at the real TCGA shape (node_num 5135 x 3 omics,
25 015 gene memberships, 438 segments) with config/kirc.yaml- and config/gbm.yaml-like flags on
synthetic data (the dataset does not ship with the reference), HIP path vs CPU oracle, 1e-4."""
from types import SimpleNamespace

import pytest
import torch

from _util import assert_close, make_args
from oracle import models as M

pytestmark = pytest.mark.gpu

KIRC = dict(model="multilevel_gnn", num_layers=2, hidden_channels=64, final_channels=32, final_head=4,
            node_embedding=True, node_embedding_dim=32, gnn_name="sage", head_dim=512, use_age=False,
            weighted_edge=True, value_att_mask=True, pca_match_mask=True, mutual_info_mask=True,
            learnable_pca=True, pca_indep_loss=True, pca_dim=3, pathway_pool_dim=1, pca_pool_dim=1,
            feature_drop=False, dropout=0.0)
GBM = dict(KIRC, node_embedding_dim=64, head_dim=256, use_age=True, pca_dim=2, pathway_pool_dim=4, pca_pool_dim=2)


def _synthetic_tcga(B, gen, n_edges=60000):
    NN, G, S = 5135 * 3, 25015, 438
    # one shared topology per fold (multiloader.py:687-691), weights in [-1, 1] incl. cross-omics +-1
    src = torch.randint(0, NN, (n_edges,), generator=gen)
    dst = torch.randint(0, NN, (n_edges,), generator=gen)
    w = torch.rand(n_edges, 1, generator=gen) * 2 - 1
    w[:2000] = torch.where(torch.rand(2000, 1, generator=gen) < 0.5, -1.0, 1.0)
    ei = torch.cat([torch.stack([src, dst]) + b * NN for b in range(B)], dim=1)
    seg = torch.sort(torch.randint(0, S, (G,), generator=gen))[0]
    match = torch.randint(0, NN, (G,), generator=gen)
    match[torch.rand(G, generator=gen) < 0.02] = -1
    return SimpleNamespace(x=torch.rand(B * NN, 1, generator=gen), edge_index=ei, edge_attr=w.repeat(B, 1),
                           gene_pca_match=match[None].repeat(B, 1), raw_indice=seg[None].repeat(B, 1),
                           age=torch.rand(B, generator=gen)), seg


@pytest.mark.parametrize("name,cfg", [("kirc", KIRC), ("gbm", GBM)])
def test_multilevel_gnn_tcga_shape(name, cfg):
    from models import get_model
    gen = torch.Generator().manual_seed(42)
    B = 2
    torch.manual_seed(7)
    args = make_args(**cfg)
    model = get_model("multilevel_gnn")(args)
    mask = (torch.rand(25015, generator=gen) > 0.3).to(torch.float32)
    model.set_pca_params(torch.randn(int(mask.sum()), args.pca_dim, generator=gen) * 0.1, mask)
    model.set_info_mask(mask[:, None].clone())
    batch, seg = _synthetic_tcga(B, gen)
    model.set_pathway_indexs(seg)
    model.eval()
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k != "info_mask")
          for k, v in model.state_dict().items()}
    pred_ref, feat_ref = M.multilevel_gnn_forward(args, sd, batch, 5135)
    floss_ref = M.feature_loss(args, sd, feat_ref, seg)
    cot = torch.randn(B, 2, generator=gen)
    names = [k for k, v in sd.items() if v.requires_grad]
    g_ref = dict(zip(names, torch.autograd.grad((pred_ref * cot).sum() + floss_ref, [sd[k] for k in names],
                                                allow_unused=True)))

    dev = "cuda:0"
    model.to(dev)
    model.set_pathway_indexs(seg.to(dev))
    gb = SimpleNamespace(**{k: v.to(dev) for k, v in vars(batch).items()})
    pred, feat = model(gb)
    assert tuple(feat.shape) == (B, 32, 146, 3 * args.pca_dim)
    # RELATIVE 1e-4 on pca_feature / pred, entry by entry
    assert_close(feat, feat_ref, 1e-4, name + " pca_feature", elementwise=True)
    assert_close(pred, pred_ref, 1e-4, name + " pred", elementwise=True)
    floss = model.get_feature_loss(feat)
    assert_close(floss, floss_ref, 1e-4, name + " feature loss")
    ((pred * cot.to(dev)).sum() + floss).backward()
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ref = g_ref[k] if g_ref[k] is not None else torch.zeros_like(sd[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        # the per-node embedding scale is an input-side gradient ([15 405, dim], one row per node): entry by entry
        assert_close(got, ref, 1e-4, name + " grad " + k, elementwise=(k == "node_embedding"))


def test_kirc_batch_of_64_properties():
    """config/kirc.yaml's batch size (64 graphs x 15 405 nodes; the CPU oracle needs minutes per sample at this size):
    size-independent properties instead -- every sample of the batch equals the same sample run alone (outputs AND the
    gradient it sends to the shared node embedding adds up), two runs are bitwise equal, all gradients are finite."""
    from models import get_model
    gen = torch.Generator().manual_seed(43)
    B, dev = 64, "cuda:0"
    torch.manual_seed(8)
    args = make_args(**KIRC)
    model = get_model("multilevel_gnn")(args)
    mask = (torch.rand(25015, generator=gen) > 0.3).to(torch.float32)
    model.set_pca_params(torch.randn(int(mask.sum()), args.pca_dim, generator=gen) * 0.1, mask)
    model.set_info_mask(mask[:, None].clone())
    batch, seg = _synthetic_tcga(B, gen)
    model.to(dev).eval()
    model.set_pathway_indexs(seg.to(dev))
    gb = SimpleNamespace(**{k: v.to(dev) for k, v in vars(batch).items()})
    cot = torch.randn(B, 2, generator=gen).to(dev)

    def run(b):
        for p in model.parameters():
            p.grad = None
        pred, feat = model(b)
        ((pred * cot[:pred.shape[0]]).sum() + model.get_feature_loss(feat)).backward()
        return pred.detach().clone(), feat.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()
                                                              if p.grad is not None}

    pred, feat, grads = run(gb)
    assert tuple(feat.shape) == (B, 32, 146, 3 * args.pca_dim) and tuple(pred.shape) == (B, 2)
    assert all(bool(torch.isfinite(g).all()) for g in grads.values()) and bool(torch.isfinite(pred).all())
    assert float(grads["node_embedding"].abs().max()) > 0
    pred2, feat2, grads2 = run(gb)                                   # determinism: no atomics on the path
    assert torch.equal(pred, pred2) and torch.equal(feat, feat2)
    assert all(torch.equal(grads[k], grads2[k]) for k in grads)
    NN = 5135 * 3
    for i in (7, 63):                                                # batch independence
        one = SimpleNamespace(x=gb.x[i * NN:(i + 1) * NN], edge_index=gb.edge_index[:, :60000],
                              edge_attr=gb.edge_attr[:60000], gene_pca_match=gb.gene_pca_match[:1],
                              raw_indice=gb.raw_indice[:1], age=gb.age[i:i + 1])
        p1, f1 = model(one)
        assert_close(f1[0], feat[i], 1e-5, "sample %d alone: pca_feature" % i, elementwise=True)
        assert_close(p1[0], pred[i], 1e-5, "sample %d alone: pred" % i, elementwise=True)
