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
