"""Host-side logic that needs no GPU: CSR construction, self-loop rewrite, state_dict surface."""
from types import SimpleNamespace

import torch

from _util import golden_files, literal, load_golden, make_args
from oracle import primitives as P


def test_csr_graph_layout():
    from mlgnn import CSRGraph
    #            e0 e1 e2 e3 e4
    src = torch.tensor([2, 0, 2, 1, 0])
    dst = torch.tensor([1, 1, 0, 1, 2])
    g = CSRGraph(torch.stack([src, dst]), 4)
    assert g.rowptr.tolist() == [0, 1, 4, 5, 5]
    assert g.col.tolist() == [2, 2, 0, 1, 0]            # row 1 keeps COO order e0, e1, e3
    assert g.eid.tolist() == [2, 0, 1, 3, 4]
    assert g.rowptr_t.tolist() == [0, 2, 3, 5, 5]
    assert g.col_t.tolist() == [1, 2, 1, 0, 1]
    assert g.pos_t.tolist() == [2, 4, 3, 0, 1]
    assert g.eid_t.tolist() == [1, 4, 3, 2, 0]
    assert g.in_degree.tolist() == [1.0, 3.0, 1.0, 0.0]
    a = torch.tensor([10., 11., 12., 13., 14.])
    by_dst, by_src = g.edge_scalar(a)
    assert by_dst.tolist() == [12., 10., 11., 13., 14.]
    assert by_src.tolist() == [11., 14., 13., 12., 10.]
    # every (src, dst) pair survives both orderings
    pairs = sorted(zip(src.tolist(), dst.tolist()))
    rows = torch.repeat_interleave(torch.arange(4), (g.rowptr[1:] - g.rowptr[:-1]).long())
    assert sorted(zip(g.col.tolist(), rows.tolist())) == pairs
    rows_t = torch.repeat_interleave(torch.arange(4), (g.rowptr_t[1:] - g.rowptr_t[:-1]).long())
    assert sorted(zip(rows_t.tolist(), g.col_t.tolist())) == pairs


def test_edge_caches_hit_for_equal_views_and_miss_after_an_edit():
    """Two views of the same elements (``a[:, 0]`` taken twice: the bench pre-builds the table from one, the model looks
    it up with another) share one cache entry; an in-place edit or a different slice does not."""
    from mlgnn import CSRGraph
    g = CSRGraph(torch.tensor([[2, 0, 2, 1, 0], [1, 1, 0, 1, 2]]), 4)
    attr = torch.arange(10.).reshape(5, 2)
    t1 = g.edge_table(attr[:, 0].reshape(-1, 1), 1)
    t2 = g.edge_table(attr[:, 0].reshape(-1, 1), 1)
    assert t1[0] is t2[0] and t1[1] is t2[1]
    other = g.edge_table(attr[:, 1].reshape(-1, 1), 1)           # same storage, other offset
    assert other[0] is not t1[0] and other[0][:, 0].tolist() == [5., 1., 3., 7., 9.]
    s1 = g.edge_scalar(attr[:, 0])
    assert g.edge_scalar(attr[:, 0])[0] is s1[0]
    attr[0, 0] = 100.                                            # version bump: both caches must rebuild
    assert g.edge_scalar(attr[:, 0])[0] is not s1[0]
    t3 = g.edge_table(attr[:, 0].reshape(-1, 1), 1)
    assert t3[0][:, 0].tolist() == [4., 100., 2., 6., 8.]
    assert g.edge_table(attr[:, 0].clone().reshape(-1, 1), 1)[0] is not t3[0]      # equal values, other storage


def test_sage_graph_matches_self_loop_rewrite():
    from mlgnn.graph import sage_graph
    ei = torch.tensor([[0, 1, 2, 2, 1], [1, 1, 0, 2, 0]])
    ea = torch.tensor([[.1], [.2], [.3], [.4], [-.5]])
    g, w = sage_graph(ei, ea, 3)
    g2, w2 = sage_graph(ei, ea, 3)
    assert g2 is g                                              # same tensors -> cached
    ei_ref, ea_ref = P.add_self_loops(*P.remove_self_loops(ei, ea), 1.0, 3)
    assert g.num_edges == ei_ref.shape[1] == 6
    by_dst, _ = g.edge_scalar(w)
    order = torch.sort(ei_ref[1], stable=True).indices
    assert g.col.tolist() == ei_ref[0][order].tolist()
    assert torch.allclose(by_dst, ea_ref[order, 0])


DEEPER_BASE = dict(num_layers=3, hidden_channels=32, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                   use_column="w", global_edge="none", graph_pooling="mean", norm="layer", mlp_layers=2,
                   block="res+", pathway_global_node=False, node_embedding=False, use_age=False,
                   num_layer_head=1, pathway_num=8, pathway_readout=None)


def test_state_dict_keys_match_reference():
    """Reference checkpoints must load with strict=True (keys and shapes from the golden fixtures)."""
    from models import DiffPool, get_model
    for p in golden_files("deepergcn"):
        f = load_golden(p)
        m = get_model("deepergcn")(make_args(**dict(DEEPER_BASE, **literal(f["over"]))))
        m.load_state_dict(f["sd"], strict=True)
    for p in golden_files("diffpool"):
        f = load_golden(p)
        Bp, C, hid, outc, nl, apl = [int(v) for v in f["cfg"]]
        dp = DiffPool(C, None, 146, nl, hid, outc, SimpleNamespace(pooling_type="correlation", after_pooling_layer=apl))
        dp.load_state_dict({k: torch.as_tensor(v) for k, v in f["sd"].items()}, strict=True)
    for p in golden_files("multilevel"):
        f = load_golden(p)
        m = get_model("multilevel_gnn")(make_args(**literal(f["over"])))
        assert m.node_embedding.shape[0] == 5135 * 3 and m.learnable_pca_params.shape[0] == 25015
        m.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
        m.set_pca_params(torch.zeros(int((f["sd"]["info_mask"] > 0).sum()), m.pca_dim), f["sd"]["info_mask"][:, 0])
        m.set_info_mask(f["sd"]["info_mask"].clone())
        m.load_state_dict(f["sd"], strict=True)
    for p in golden_files("mlgseq"):
        f = load_golden(p)
        m = get_model("multilevel_gnn_seq")(make_args(**literal(f["over"])))
        m.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
        m.load_ckpt({k: torch.as_tensor(v) for k, v in f["sd"].items()})
        assert sorted(m.state_dict()) == sorted(f["sd"])                  # pathwayhead.* keys, no conv_model.* / head.*
        m.load_state_dict(f["sd"], strict=True)


def test_vae_surface_and_head_sizing():
    """VAE state_dict keys / shapes equal the reference's (strict load of its fixtures), ``reconstruct_head`` sizes the
    head for ceil(146 * 0.25^L) DiffPool clusters (vae.py:278-282), the similarity matrix gets its self loops."""
    import numpy as np
    from models import get_model
    from models.vae import next_power_of_two
    assert [next_power_of_two(n) for n in (1, 2, 3, 4, 5, 17, 64)] == [1, 2, 4, 4, 8, 32, 64]
    for p in golden_files("vae"):
        f = load_golden(p)
        args = make_args(**literal(f["over"]))
        m = get_model("vae")(args, None, f["pathway_indexs"])
        assert get_model("mmd_vae") is get_model("vae")
        m.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
        m.set_pca_params(torch.zeros(int((f["sd"]["info_mask"] > 0).sum()), m.pca_dim), f["sd"]["info_mask"][:, 0])
        m.set_info_mask(f["sd"]["info_mask"].clone())
        m.reconstruct_head(args)
        assert sorted(m.state_dict()) == sorted(f["sd"])
        m.load_state_dict(f["sd"], strict=True)
        if args.reorder_type == "diff_pooling":
            assert m.head[0].in_features == args.diff_pooling_output_dim * 10 * 3 * args.pca_dim + int(args.use_age)
        m.set_pathway_similarity_matrix(np.zeros((146, 146)))
        assert torch.equal(m.get_pathway_adj(), torch.eye(146))
    for name, prefix in (("vq_vae", "vqvae"), ("autoencoder", "autoencoder")):
        for p in golden_files(prefix):
            f = load_golden(p)
            args = make_args(**literal(f["over"]))
            m = get_model(name)(args, None, f["pathway_indexs"])
            m.node_embedding = torch.nn.Parameter(f["sd"]["node_embedding"].clone())
            m.set_pca_params(torch.zeros(int((f["sd"]["info_mask"] > 0).sum()), m.pca_dim), f["sd"]["info_mask"][:, 0])
            m.set_info_mask(f["sd"]["info_mask"].clone())
            if name == "vq_vae":
                m.reconstruct_head(args)
            assert sorted(m.state_dict()) == sorted(f["sd"]), name
            m.load_state_dict(f["sd"], strict=True)


def test_gbm_parameter_count():
    """config/gbm.yaml shape: 2 858 279 parameters incl. the frozen info_mask."""
    from models import get_model
    a = make_args(model="multilevel_gnn", num_layers=2, hidden_channels=64, final_channels=32, final_head=4,
                  node_embedding=True, node_embedding_dim=64, gnn_name="sage", head_dim=256, use_age=True,
                  weighted_edge=True, value_att_mask=True, pca_match_mask=True, mutual_info_mask=True,
                  learnable_pca=True, pca_indep_loss=True, feature_drop=True)
    m = get_model("multilevel_gnn")(a)
    m.set_info_mask(torch.ones(25015, 1))          # a frozen nn.Parameter, counted by the reference too
    assert sum(p.numel() for p in m.parameters()) == 2858279
    keys = set(m.state_dict())
    assert {"node_embedding", "learnable_pca_params", "gnn_model.0.gconv.lin_l.weight", "gnn_model.1.gconv.lin_r.weight",
            "gnn_model.0.gconv.nn.0.weight", "gnn_model.0.gconv.nn.0.bias", "conv_model.0.weight", "conv_model.2.bias",
            "head.0.weight", "head.3.bias"} <= keys
    assert tuple(m.state_dict()["head.0.weight"].shape) == (256, 6913)
    assert tuple(m.state_dict()["gnn_model.1.gconv.nn.0.weight"].shape) == (32, 96)


def test_edge_cache_is_keyed_on_the_tensor_not_its_address():
    """A freed edge-weight tensor's storage is usually handed to the next tensor of the same size (same data_ptr,
    version 0): the per-graph cache must not return the old weights for it."""
    from mlgnn import CSRGraph
    ei = torch.tensor([[0, 1, 2, 3, 0], [1, 2, 3, 0, 2]])
    g = CSRGraph(ei, 4)
    a = torch.arange(5, dtype=torch.float32)
    first = g.edge_scalar(a)[0].clone()
    assert g.edge_scalar(a)[0] is g.edge_scalar(a)[0]                # same tensor: cached
    ptr = a.data_ptr()
    b = a                                                            # the cache entry keeps the old tensor alive,
    del a                                                            # so a new tensor cannot reuse its address
    c = torch.full((5,), 7.0)
    assert c.data_ptr() != ptr or c is b
    assert torch.equal(g.edge_scalar(c)[0], torch.full((5,), 7.0))
    assert not torch.equal(first, g.edge_scalar(c)[0])
    t = torch.arange(10, dtype=torch.float32).view(5, 2)
    by_dst, by_src = g.edge_table(t, 2)
    assert g.edge_table(t, 2)[0] is by_dst
    t.add_(1.0)                                                      # in-place update bumps the version
    assert torch.equal(g.edge_table(t, 2)[0], by_dst + 1.0)


def test_step_lr():
    from mlgnn.optim import StepLR

    class _Opt:
        param_groups = [dict(lr=0.1, initial_lr=0.1)]
    o = _Opt()
    sched = StepLR(o, step_size=3, gamma=0.25)
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=0.1)
    tsched = torch.optim.lr_scheduler.StepLR(topt, step_size=3, gamma=0.25)
    for _ in range(10):
        topt.step()
        tsched.step()
        sched.step()
        assert abs(sched.get_last_lr()[0] - tsched.get_last_lr()[0]) < 1e-12


def test_head_conv_1x1_equals_conv2d():
    """The pathway head's 1x1 convolution as a GEMM over the channel-last view: same parameters / state_dict keys as
    nn.Conv2d, same outputs and gradients; other kernel sizes go through nn.Conv2d itself."""
    import torch
    from models.multilevel_gnn import HeadConv2d
    torch.manual_seed(0)
    c, r = HeadConv2d(16, 8, 1, padding=0), torch.nn.Conv2d(16, 8, 1)
    assert list(c.state_dict()) == list(r.state_dict())
    r.load_state_dict(c.state_dict())
    x = torch.randn(3, 16, 5, 7, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    y, y2 = c(x), r(x2)
    assert y.shape == y2.shape and torch.allclose(y, y2, atol=1e-6)
    (y * torch.arange(7.0)).sum().backward()
    (y2 * torch.arange(7.0)).sum().backward()
    assert torch.allclose(x.grad, x2.grad, atol=1e-6) and torch.allclose(c.weight.grad, r.weight.grad, atol=1e-4)
    assert torch.allclose(c.bias.grad, r.bias.grad, atol=1e-4)
    c3 = HeadConv2d(4, 4, 3, padding=1)
    assert torch.equal(c3(x[:, :4]), torch.nn.functional.conv2d(x[:, :4], c3.weight, c3.bias, padding=1))


def test_topology_cache_identity_view_and_content():
    """CSRGraph.from_cache: the same tensor (or an equal view) hits without any work, an in-place edit misses, an equal
    COPY hits only with content=True (dataloader/multiloader.py:687-691: one topology per fold)."""
    from mlgnn import CSRGraph
    CSRGraph.clear_cache()
    gen = torch.Generator().manual_seed(3)
    ei = torch.randint(0, 50, (2, 400), generator=gen)
    g0 = CSRGraph.from_cache(ei, 50)
    assert CSRGraph.from_cache(ei, 50) is g0
    assert CSRGraph.from_cache(ei[:, :], 50) is g0                       # another view of the same elements
    assert CSRGraph.from_cache(ei, 51) is not g0                         # a different node count is a different graph
    copy = ei.clone()
    g1 = CSRGraph.from_cache(copy, 50)
    assert g1 is not g0 and torch.equal(g1.col, g0.col)                  # identity cache: an equal copy is built again
    CSRGraph.clear_cache()
    g2 = CSRGraph.from_cache(ei, 50, content=True)
    assert CSRGraph.from_cache(ei.clone(), 50, content=True) is g2       # equal contents, different tensor
    other = ei.clone()
    other[0, 7] = (other[0, 7] + 1) % 50
    assert CSRGraph.from_cache(other, 50, content=True) is not g2
    ei[1, 3] = (ei[1, 3] + 1) % 50                                       # in-place edit: the version moved on
    assert CSRGraph.from_cache(ei, 50) is not g2
    CSRGraph.clear_cache()


def test_membership_cache_hits_for_views_of_one_table():
    """Every batch of a fold views ONE membership table: the grouped index tables are built once."""
    from mlgnn import project
    del project._MEMBERSHIP_CACHE[:]
    gen = torch.Generator().manual_seed(5)
    match = torch.randint(0, 30, (200,), generator=gen)
    seg = torch.sort(torch.randint(0, 12, (200,), generator=gen))[0]
    a = project.membership_tables(match[None, :].expand(4, -1), seg[None, :].expand(4, -1), 30, 12, 120)
    b = project.membership_tables(match[None, :].expand(4, -1), seg[None, :].expand(4, -1), 30, 12, 120)
    assert a is b
    c = project.membership_tables(match[None, :].repeat(4, 1), seg[None, :].repeat(4, 1), 30, 12, 120)
    assert c is not a and torch.equal(c.seg_mem, a.seg_mem) and torch.equal(c.node_ptr, a.node_ptr)
