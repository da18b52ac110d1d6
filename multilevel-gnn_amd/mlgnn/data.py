"""Batching with the reference's data contract, without torch_geometric.

The reference batches per-patient ``torch_geometric.data.Data`` objects with PyG's ``DataLoader``
(``train.py:17,316-327``); the fields are produced by ``dataloader/multiloader.py:76-94,687-698,
1043-1052``.  This module reproduces the PyG 2.2.0 collate rules those fields rely on:

* tensors are concatenated along dim 0, except keys containing ``"index"`` (or ``"face"``), which are
  concatenated along the LAST dim and incremented by the cumulative node count of the graphs before;
* python numbers become a 1-D tensor with one entry per graph;
* ``batch`` (graph id per node), ``ptr`` (node offsets) and ``num_graphs`` are added.

``gene_pca_match`` / ``raw_indice`` lack "index" in their names and are therefore NOT offset -- the
model offsets them itself (``multilevel_gnn.py:212``).  The collate step can also pre-sort the
topology (``with_csr=True`` attaches ``batch.csr``, a host-built :class:`mlgnn.CSRGraph`).
"""
import torch
from torch.utils.data import DataLoader as _TorchLoader


class Data:
    """Attribute bag for one graph (the subset of ``torch_geometric.data.Data`` the models touch)."""

    def __init__(self, **fields):
        for k, v in fields.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in vars(self) if not k.startswith("_")]

    @property
    def num_nodes(self):
        x = getattr(self, "x", None)
        if x is not None:
            return x.shape[0]
        return int(self.edge_index.max()) + 1 if self.edge_index.numel() else 0

    def to(self, device, non_blocking=False):
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
            elif hasattr(v, "to") and k == "csr":
                setattr(self, k, v.to(device))
            elif hasattr(v, "to") and k == "shared_topology":
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self


class Batch(Data):
    @staticmethod
    def _cat_dim(key):
        return -1 if ("index" in key or key == "face") else 0

    @staticmethod
    def _increments(key):
        return ("index" in key or key == "face") and key != "batch"

    @classmethod
    def from_data_list(cls, data_list, with_csr=False):
        if not data_list:
            raise ValueError("empty batch")
        out = cls()
        sizes = [d.num_nodes for d in data_list]
        offsets = [0]
        for n in sizes:
            offsets.append(offsets[-1] + n)
        for key in data_list[0].keys():
            vals = [getattr(d, key) for d in data_list]
            v0 = vals[0]
            if torch.is_tensor(v0):
                if cls._increments(key):
                    vals = [v + off for v, off in zip(vals, offsets)]
                if v0.dim() == 0:
                    setattr(out, key, torch.stack(vals))
                else:
                    setattr(out, key, torch.cat(vals, dim=cls._cat_dim(key)))
            elif isinstance(v0, (int, float, bool)):
                setattr(out, key, torch.tensor(vals))
            else:
                setattr(out, key, vals)
        out.batch = torch.repeat_interleave(torch.arange(len(data_list)), torch.tensor(sizes))
        out.ptr = torch.tensor(offsets)
        out.num_graphs = len(data_list)
        shared = cls._shared_topology(data_list, sizes)
        if shared is not None:
            out.shared_topology = shared
        if with_csr:
            from .graph import CSRGraph
            out.csr = CSRGraph(out.edge_index, offsets[-1])
        return out


    # ONE topology for every sample (dataloader/multiloader.py:687-691 assigns the same edge list to every patient of a
    # fold): noticed here, on the loader side, so that the model can take the per-fold CSR instead of sorting the B-fold
    # edge list of every batch.  Same tensor object in all samples: free; equal contents in different tensors (the
    # reference builds one tensor per patient): compared once per pair of tensor identities.
    _EQUAL = {}

    @classmethod
    def _same(cls, a, b):
        if a is b:
            return True
        if a is None or b is None or a.shape != b.shape or a.dtype != b.dtype:
            return False
        key = (id(a), a._version, id(b), b._version)
        hit = cls._EQUAL.get(key)
        if hit is None or hit[1]() is not a or hit[2]() is not b:            # (ids can be recycled: weak references pin them)
            import weakref
            if len(cls._EQUAL) > 4096:
                cls._EQUAL.clear()
            hit = cls._EQUAL[key] = (bool(torch.equal(a, b)), weakref.ref(a), weakref.ref(b))
        return hit[0]

    @classmethod
    def _shared_topology(cls, data_list, sizes):
        d0 = data_list[0]
        ei0, ea0 = getattr(d0, "edge_index", None), getattr(d0, "edge_attr", None)
        if not torch.is_tensor(ei0) or ei0.dim() != 2 or len(set(sizes)) != 1 or ei0.shape[1] == 0:
            return None
        for d in data_list[1:]:
            if not cls._same(ei0, getattr(d, "edge_index", None)):
                return None
            ea = getattr(d, "edge_attr", None)
            if (ea0 is None) != (ea is None) or (ea0 is not None and not cls._same(ea0, ea)):
                return None
        from .graph import SharedTopology
        return SharedTopology(ei0, ea0 if torch.is_tensor(ea0) else None, sizes[0], len(data_list))


class DataLoader(_TorchLoader):
    """``torch_geometric.data.DataLoader`` stand-in: same constructor keywords the reference uses
    (``batch_size, shuffle, num_workers, drop_last``), PyG collate rules."""

    def __init__(self, dataset, batch_size=1, shuffle=False, with_csr=False, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle,
                         collate_fn=lambda items: Batch.from_data_list(items, with_csr=with_csr), **kwargs)


class SyntheticTCGA(torch.utils.data.Dataset):
    """Patients with the shape of the reference's dataset (which does not ship with it): 3 omics x
    ``node_num`` gene nodes with one scalar value each, ONE topology shared by all patients
    (``multiloader.py:687-691``) with weighted edges incl. -1/+1 cross-omics edges (:664-671), a
    sorted gene -> (pathway, omics) membership table of ``n_members`` entries over 146 x 3 segments,
    a binary label that depends on a few pathway nodes (so that training can reduce the loss)."""

    def __init__(self, n_patients, node_num=5135, n_edges=60000, n_members=25015, pca_dim=2, seed=0):
        gen = torch.Generator().manual_seed(seed)
        self.node_num, self.NN = node_num, node_num * 3
        NN = self.NN
        src = torch.randint(0, NN, (n_edges,), generator=gen)
        dst = torch.randint(0, NN, (n_edges,), generator=gen)
        w = torch.rand(n_edges, 1, generator=gen)
        n_cross = min(n_edges // 10, node_num)
        src[:n_cross] = torch.arange(n_cross)
        dst[:n_cross] = torch.arange(n_cross) + node_num                      # CNV -> mRNA of the same gene
        w[:n_cross] = torch.where(torch.rand(n_cross, 1, generator=gen) < 0.5, -1.0, 1.0)
        self.edge_index, self.edge_attr = torch.stack([src, dst]), w
        self.raw_indice = torch.sort(torch.randint(0, 438, (n_members,), generator=gen))[0]
        self.gene_pca_match = torch.randint(0, NN, (n_members,), generator=gen)
        self.gene_pca_match[torch.rand(n_members, generator=gen) < 0.02] = -1
        self.n_members, self.pca_dim = n_members, pca_dim
        self.x = torch.rand(n_patients, NN, 1, generator=gen)
        signal = self.x[:, self.gene_pca_match.clamp(min=0)[:200], 0].mean(1)
        self.labels = (signal > signal.median()).long()
        self.age = torch.rand(n_patients, generator=gen)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        y = torch.tensor([1.0 - float(self.labels[i]), float(self.labels[i])])
        return Data(x=self.x[i], edge_index=self.edge_index, edge_attr=self.edge_attr, y=y, age=float(self.age[i]),
                    gene_pca_match=self.gene_pca_match[None, :], raw_indice=self.raw_indice[None, :],
                    node_size=self.NN, pathway_node_attr=torch.zeros(1, 146, 3 * self.pca_dim))

    def get_weight_balance(self, indexs, batch_size, weight_power=1.0):
        """``MyData.get_weight_balance`` (multiloader.py:321-326): per-class weights repeated per batch row."""
        counts = torch.bincount(self.labels[torch.as_tensor(indexs)], minlength=2).float()
        return torch.repeat_interleave(((counts.max() / counts) ** weight_power).unsqueeze(0), batch_size, dim=0)
