"""PyG-compatible collate rules and the synthetic TCGA-shaped dataset (host side, no GPU)."""
import torch


def _sample(n, e, seed):
    from mlgnn.data import Data
    g = torch.Generator().manual_seed(seed)
    return Data(x=torch.rand(n, 1, generator=g), edge_index=torch.randint(0, n, (2, e), generator=g),
                edge_attr=torch.rand(e, 1, generator=g), y=torch.tensor([1.0, 0.0]), age=0.5 + seed,
                gene_pca_match=torch.randint(0, n, (1, 7), generator=g), raw_indice=torch.arange(7)[None, :],
                node_size=n, pathway_node_attr=torch.zeros(1, 146, 6))


def test_collate_follows_pyg_rules():
    from mlgnn.data import Batch
    a, b, c = _sample(5, 4, 0), _sample(3, 6, 1), _sample(4, 2, 2)
    batch = Batch.from_data_list([a, b, c])
    assert batch.num_graphs == 3 and batch.ptr.tolist() == [0, 5, 8, 12]
    assert batch.x.shape == (12, 1) and batch.edge_attr.shape == (12, 1)
    # keys containing "index": concatenated on the last dim and offset by the node count before
    assert batch.edge_index.shape == (2, 12)
    assert torch.equal(batch.edge_index[:, 4:10], b.edge_index + 5)
    assert torch.equal(batch.edge_index[:, 10:], c.edge_index + 8)
    # no "index" in the name: concatenated on dim 0, NOT offset (the model offsets it, multilevel_gnn.py:212)
    assert batch.gene_pca_match.shape == (3, 7) and torch.equal(batch.gene_pca_match[1], b.gene_pca_match[0])
    assert batch.raw_indice.shape == (3, 7)
    assert batch.y.shape == (6,) and batch.y.reshape(-1, 2).shape == (3, 2)
    assert batch.age.tolist() == [0.5, 1.5, 2.5] and batch.node_size.tolist() == [5, 3, 4]
    assert batch.pathway_node_attr.shape == (3, 146, 6)
    assert batch.batch.tolist() == [0] * 5 + [1] * 3 + [2] * 4


def test_loader_and_presorted_topology():
    from mlgnn import CSRGraph
    from mlgnn.data import DataLoader, SyntheticTCGA
    data = SyntheticTCGA(10, node_num=20, n_edges=90, n_members=50, seed=3)
    loader = DataLoader(data, batch_size=4, shuffle=False, drop_last=True, with_csr=True)
    batches = list(loader)
    assert len(batches) == 2
    b = batches[0]
    assert b.x.shape == (4 * 60, 1) and b.edge_index.shape == (2, 4 * 90) and b.y.shape == (8,)
    assert int(b.edge_index.max()) < 240 and b.gene_pca_match.shape == (4, 50)
    ref = CSRGraph(b.edge_index, 240)
    assert torch.equal(ref.col, b.csr.col) and torch.equal(ref.rowptr_t, b.csr.rowptr_t)
    w = data.get_weight_balance(list(range(10)), 4)
    assert w.shape == (4, 2) and float(w.min()) == 1.0


def test_harness_options_follow_the_reference_yaml():
    import os
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    import train_harness as th
    a = th.parse_opts([])
    assert a.lr == 1e-4 and a.batch_size == 4 and a.gcn_aggr == "max" and a.conv_channel_list == [32, 64]
    cfg = os.path.join(os.path.dirname(__file__), "golden", "gbm_like.yaml")
    a = th.parse_opts(["--config", cfg, "--epochs", "2"])
    assert a.epochs == 2                               # explicit CLI flag kept
    assert a.batch_size == 32 and a.gnn_name == "sage" and a.hidden_channels == 64 and a.weight_balance is True


def test_collate_notices_a_topology_shared_by_all_samples():
    """The reference's loader gives every patient of a fold the same edge list (dataloader/multiloader.py:687-691): the
    collate attaches ``shared_topology`` (one sample's edge list, nodes per sample, copies) -- whether the samples hold the
    same tensor object or equal tensors of their own -- and does not when any sample differs."""
    from mlgnn.data import Batch, SyntheticTCGA
    ds = SyntheticTCGA(6, node_num=40, n_edges=300, n_members=200)
    items = [ds[i] for i in range(4)]
    b = Batch.from_data_list(items)
    st = b.shared_topology
    assert st.copies == 4 and st.nodes == 120 and st.edge_index is ds.edge_index and st.edge_attr is ds.edge_attr
    assert torch.equal(b.edge_index, torch.cat([ds.edge_index + 120 * k for k in range(4)], dim=1))
    for it in items:                                        # equal contents in tensors of their own (the reference's way)
        it.edge_index, it.edge_attr = it.edge_index.clone(), it.edge_attr.clone()
    assert Batch.from_data_list(items).shared_topology.copies == 4
    items[2].edge_index = items[2].edge_index.clone()
    items[2].edge_index[0, 5] += 1
    assert not hasattr(Batch.from_data_list(items), "shared_topology")
    items[2] = ds[2]
    items[1].edge_attr = items[1].edge_attr + 1.0
    assert not hasattr(Batch.from_data_list(items), "shared_topology")
