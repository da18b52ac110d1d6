"""Device COO->CSR build (counting sort + per-row bitonic fix-up) vs the host construction: bit-exact
index arrays, whatever order the atomics hand out slots in."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ("rowptr", "col", "eid", "rowptr_t", "col_t", "pos_t", "eid_t")


@pytest.mark.parametrize("N,E", [(1, 0), (5, 0), (1, 3), (2, 1), (7, 50), (1000, 20000), (100003, 777777),
                                 (640000, 2000000)])
def test_device_csr_matches_host(N, E):
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(N + E)
    ei = torch.randint(0, N, (2, E), generator=gen)
    if E > 100:
        ei[1, :50] = N - 1            # a heavy last row, leading rows possibly empty
        ei[0, 50:90] = 0
    host = CSRGraph(ei, N)
    dev = CSRGraph(ei.to("cuda:0"), N)
    for f in FIELDS:
        a, b = getattr(host, f), getattr(dev, f).cpu()
        assert a.dtype == b.dtype == torch.int32 and a.shape == b.shape, f
        assert torch.equal(a, b), f


def test_headline_batch_shape_matches_host():
    """64 graphs x 10 000 nodes / 160 000 edges as one block-diagonal edge list (BASELINE configs[1]): the chunk length
    of the count / placement workgroups is at its cap here (20 edges per thread, ~500 workgroups in XCD-contiguous runs)."""
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(4)
    B, n, e = 64, 10000, 160000
    ei = torch.randint(0, n, (2, B, e), generator=gen) + (torch.arange(B) * n)[None, :, None]
    ei = ei.reshape(2, B * e)
    host = CSRGraph(ei, B * n)
    dev = CSRGraph(ei.to("cuda:0"), B * n)
    for f in FIELDS:
        assert torch.equal(getattr(host, f), getattr(dev, f).cpu()), f


@pytest.mark.parametrize("N,E,hub_in,hub_out", [(3000, 60000, 33, 64), (3000, 60000, 65, 700), (2000, 90000, 4096, 4097),
                                               (500, 70000, 20000, 15000), (64, 30000, 0, 0)])
def test_long_rows(N, E, hub_in, hub_out):
    """Rows of 33..64 edges (one row per wavefront), 65..4096 (LDS network) and beyond (global-memory network),
    on both the destination and the source side; a dense small graph where every row is long; run twice:
    the result must not depend on the arrival order of the atomics."""
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(E + hub_in)
    ei = torch.randint(0, N, (2, E), generator=gen)
    ei[1, :hub_in] = 7                                    # destination hub
    ei[0, E - hub_out:] = 11                              # source hub
    host = CSRGraph(ei, N)
    for _ in range(2):
        dev = CSRGraph(ei.to("cuda:0"), N)
        for f in FIELDS:
            assert torch.equal(getattr(host, f), getattr(dev, f).cpu()), f


def test_int32_edge_index_and_noncontiguous_input():
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(3)
    ei = torch.randint(0, 50, (300, 2), generator=gen).t()           # non-contiguous view
    host = CSRGraph(ei.contiguous(), 50)
    dev = CSRGraph(ei.to("cuda:0").to(torch.int32), 50)
    for f in FIELDS:
        assert torch.equal(getattr(host, f), getattr(dev, f).cpu()), f


def test_out_of_range_ids_are_clamped_and_reported():
    from mlgnn import CSRGraph, gen_aggregate
    ei = torch.tensor([[0, 1, 7, 2], [1, -3, 2, 0]], device="cuda:0")        # 7 and -3 are not nodes of a 4-node graph
    g = CSRGraph(ei, 4)
    out = gen_aggregate(torch.ones(4, 8, device="cuda:0"), g, None, aggr="add")   # must not fault
    assert bool(torch.isfinite(out).all())
    with pytest.raises(ValueError, match="outside"):
        g.validate()
    CSRGraph(torch.tensor([[0, 1], [1, 0]], device="cuda:0"), 2).validate()


@pytest.mark.parametrize("r,width", [(1, 1), (7, 8), (2, 2), (3, 4)])
def test_edge_table_device_gather_matches_host(r, width):
    """Raw edge attributes -> by-destination / by-source order, zero padded: device kernel vs indexing."""
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(r)
    N, E = 500, 6000
    ei = torch.randint(0, N, (2, E), generator=gen)
    attr = torch.rand(E, r, generator=gen)
    g = CSRGraph(ei.cuda(), N)
    by_dst, by_src = g.edge_table(attr.cuda(), width)
    ref_dst = torch.nn.functional.pad(attr, (0, width - r))[g.eid.cpu().long()]
    ref_src = ref_dst[g.pos_t.cpu().long()]
    assert torch.equal(by_dst.cpu(), ref_dst) and torch.equal(by_src.cpu(), ref_src)
    # a strided column view of a wider table (the use_column case) and a float64 table
    wide = torch.rand(E, 5, generator=gen).cuda()
    d2, s2 = g.edge_table(wide[:, 2:3], 1)
    assert torch.equal(d2.cpu()[:, 0], wide.cpu()[:, 2][g.eid.cpu().long()])
    assert torch.equal(s2.cpu()[:, 0], wide.cpu()[:, 2][g.eid_t.cpu().long()])
    d3, _ = g.edge_table(attr.double().cuda(), width)
    assert torch.equal(d3.cpu(), ref_dst)


def test_edge_table_one_column_tail():
    """One column, an edge count that is not a multiple of four: the 16-byte path plus its scalar tail."""
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(2)
    N, E = 300, 4099
    ei = torch.randint(0, N, (2, E), generator=gen)
    attr = torch.rand(E, 1, generator=gen)
    g = CSRGraph(ei.cuda(), N)
    by_dst, by_src = g.edge_table(attr.cuda(), 1)
    assert torch.equal(by_dst.cpu(), attr[g.eid.cpu().long()])
    assert torch.equal(by_src.cpu(), attr[g.eid_t.cpu().long()])


def test_replicated_graph_is_the_graph_of_the_batched_edge_list():
    """``CSRGraph.replicated(single, B)`` (mlgnn_csr_replicate) against the CSR built from the B-fold block-diagonal edge
    list: every array bit for bit (both orderings are stable, copy by copy)."""
    import torch
    from mlgnn import CSRGraph
    gen = torch.Generator().manual_seed(12)
    n, e, B = 1234, 9000, 7
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)])
    ei[1, :400] = 5                                           # a long row
    dev = "cuda:0"
    single = CSRGraph(ei.to(dev), n)
    rep = CSRGraph.replicated(single, B)
    full = CSRGraph(torch.cat([ei + k * n for k in range(B)], dim=1).to(dev), n * B)
    for name in ("rowptr", "col", "eid", "rowptr_t", "col_t", "pos_t", "eid_t"):
        assert torch.equal(getattr(rep, name), getattr(full, name)), name
    assert rep.num_nodes == full.num_nodes and rep.num_edges == full.num_edges


def test_shared_topology_gives_the_same_sage_layer():
    """SAGEConv over a batch whose samples share one graph: through the per-fold CSR (``shared=``; 1 / in-degree folded
    into the cached per-edge weights) and through the sort of the batched edge list -- same outputs and gradients to fp32
    rounding; two runs through the shared path are bitwise equal; the second call with the same fold tensors does no
    topology work at all (cache hit)."""
    import torch
    from mlgnn import graph as G
    from models.gcn_lib.sparse.torch_vertex import GraphConv
    gen = torch.Generator().manual_seed(13)
    n, e, B, cin, cout = 3000, 20000, 5, 32, 64
    ei = torch.stack([torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)])
    ei[1, :30] = ei[0, :30]                                   # self loops: dropped either way
    w = torch.rand(e, 1, generator=gen) * 2 - 1
    dev = "cuda:0"
    torch.manual_seed(2)
    conv = GraphConv(cin, cout, conv="sage", act="leakyrelu", mlp_norm="none").to(dev)
    x = torch.randn(B * n, cin, generator=gen).to(dev)
    ei_b = torch.cat([ei + k * n for k in range(B)], dim=1).to(dev)
    w_b = w.repeat(B, 1).to(dev)
    cot = torch.randn(B * n, cout, generator=gen).to(dev)
    shared = G.SharedTopology(ei.to(dev), w.to(dev), n, B)

    def run(sh):
        for p in conv.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        out = conv(xg, ei_b, w_b, shared=sh)
        (out * cot).sum().backward()
        return out.detach(), xg.grad, [p.grad.clone() for p in conv.parameters() if p.grad is not None]

    from _util import assert_close
    o0, gx0, gp0 = run(None)
    o1, gx1, gp1 = run(shared)
    o2, gx2, gp2 = run(shared)
    assert torch.equal(o1, o2) and torch.equal(gx1, gx2) and all(torch.equal(a, b) for a, b in zip(gp1, gp2))
    assert_close(o1, o0, 1e-5, "out", elementwise=True)
    assert_close(gx1, gx0, 1e-5, "grad x", elementwise=True)
    for a, b in zip(gp1, gp0):
        assert_close(a, b, 1e-5, "param grad")
    g_first = G.shared_sage_graph(shared, dev)[0]
    assert G.shared_sage_graph(shared, dev)[0] is g_first      # per-fold cache: no work the second time
    assert g_first.num_nodes == B * n
