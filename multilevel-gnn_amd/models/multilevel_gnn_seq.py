"""MultilevelGNNSeq on the HIP kernels (interface of the reference's ``models/multilevel_gnn_seq.py``:
``PathwayHeadSeq`` :14-68, ``MultilevelGNNSeq`` :70-404).

The graph levels are those of :class:`models.multilevel_gnn.MultilevelGNN` (embedding scale, GraphConv stack on the
CSR kernels, value mask, gene -> pathway projection pooling); the variant only re-homes level 2: the conv stack,
max-pool, dropout and MLP head live in a ``PathwayHeadSeq`` submodule (``state_dict`` keys ``pathwayhead.*``), the
omics selection is the ``only_mrna_pred`` switch (first two columns of the conv output, no ``drop1``) instead of
``used_omics``, and ``load_ckpt`` restores a checkpoint around a re-created projection parameter.
"""
import torch
import torch.nn as nn

from .multilevel_gnn import N_OMICS, N_PATHWAYS, HeadConv2d, MultilevelGNN


class PathwayHeadSeq(nn.Module):
    """Reference :14-68.  ``forward(x [B,C,146,3k], age [B] | None) -> probabilities [B,2]``."""

    def __init__(self, args):
        super().__init__()
        if args.pca_compare:             # the reference's branch reads a ``pre_linear`` this module never creates
            raise NotImplementedError("pca_compare is outside the accelerated path")
        self.head_dim = args.head_dim
        self.pathway_pool_dim = args.pathway_pool_dim
        self.pca_pool_dim = args.pca_pool_dim
        self.pca_dim = args.pca_dim
        self.pca_compare = args.pca_compare
        self.args = args

        convs, cin = [], args.final_channels
        for cout, kern in zip(args.conv_channel_list, args.conv_kernel_list):
            convs += [HeadConv2d(cin, cout, kern, padding=kern // 2), nn.ReLU()]
            cin = cout
        self.conv_model = nn.ModuleList(convs)
        self.pooling = nn.MaxPool2d((self.pathway_pool_dim, self.pca_pool_dim))
        self.drop1 = nn.Dropout(0.25 if args.feature_drop else 0)
        width = self.pca_dim if args.only_mrna_pred else N_OMICS * self.pca_dim
        head_in = args.conv_channel_list[-1] * (N_PATHWAYS // self.pathway_pool_dim) * (width // self.pca_pool_dim) \
            + (1 if args.use_age else 0)
        self.head = nn.Sequential(nn.Linear(head_in, self.head_dim), nn.ReLU(), nn.Dropout(0.5),
                                  nn.Linear(self.head_dim, 2), nn.Softmax(dim=1))

    def forward(self, x, age=None):
        for layer in self.conv_model:
            x = layer(x)
        if not self.args.only_mrna_pred:
            x = self.pooling(x)
            x = self.drop1(x)
            x = torch.flatten(x, start_dim=1)
        else:
            x = x[:, :, :, :2]           # literal 2 in the reference (:62): the mRNA block at pca_dim = 2
            x = self.pooling(x)
            x = torch.flatten(x, start_dim=1)
        if self.args.use_age:
            x = torch.cat([x, age[:, None]], dim=-1)
        return self.head(x)


class MultilevelGNNSeq(MultilevelGNN):

    def _build_head(self, args):
        self.pathwayhead = PathwayHeadSeq(args)

    def _apply_head(self, x, age):
        return self.pathwayhead(x, age) if self.args.use_age else self.pathwayhead(x)

    def load_ckpt(self, state_dict):
        """Reference :397-404: re-create the projection parameter at the checkpoint's size, take its mask, then load
        every tensor whose name this model has."""
        self.learnable_pca_params = nn.Parameter(torch.zeros(state_dict['learnable_pca_params'].shape),
                                                 requires_grad=(not self.args.freeze_pca_weight))
        self.set_info_mask(state_dict['info_mask'])
        names = [n for n, _ in self.named_parameters()]
        self.load_state_dict({k: v for k, v in state_dict.items() if k in names}, strict=False)
