"""VAE, VQ_VAE and AutoEncoder on the HIP kernels (interfaces of the reference's ``models/vae.py``,
``models/vq_vae.py`` :36-434 and ``models/autoencoder.py`` :23-151).  ``models/vae.py``: class ``VAE(MultilevelGNN)`` :39,
``encoder`` :128-208, ``train_step`` / ``eval_step`` :90-117, ``forward`` :119-126, decoders :210-222,
``predict_head`` :233-265 -- the caller of ``DiffPool`` --, ``reconstruct_head`` :267-299,
``set_pathway_similarity_matrix`` :305, ``vae_loss`` :334-357 and the MMD kernels :376-446).

Same constructor, methods, return values and ``state_dict`` keys.  Level 0 (GraphConv stack) and the gene -> pathway
projection run on the CSR / segment kernels, the pooled levels on the fused DiffPool / DenseSAGE kernels.  Two
loops of the reference are replaced by batched forms with identical results: the per-pathway ``corrcoef`` loop of
the encoder loss (438 launches -> one batched covariance) and, for the uniform-width ``foreach`` decoder, the
438-block Python loop (-> one batched GEMM + one gathered row-dot over all output genes).
``get_embedding_similarity`` (spreadsheet ETL) is outside the accelerated path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from mlgnn.project import segment_project
from .diff_pooling import DiffPool
from .multilevel_gnn import N_OMICS, N_PATHWAYS, MultilevelGNN


def next_power_of_two(n):
    """Smallest power of two >= n (reference ``findNextPowerOf2`` :15-28)."""
    return 1 if n <= 1 else 1 << (int(n) - 1).bit_length()


def _xavier(block):
    for m in block.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.xavier_uniform_(m.weight.data)


def _head(in_dim, hidden):
    return nn.Sequential(nn.Linear(in_dim, hidden), nn.ReLU(), nn.Dropout(0.5), nn.Linear(hidden, 2), nn.Softmax(dim=1))


class _PretrainBase(MultilevelGNN):
    """What the reference's three pre-training models (autoencoder.py, vae.py, vq_vae.py) repeat verbatim: the decoder
    construction, the encoder front (GraphConv stack -> gather by gene_pca_match -> projection pooling) and the two
    decoders."""

    def __init__(self, args, pca_params=None, pathway_indexs=None):
        super().__init__(args, pca_params, pathway_indexs)
        if args.dense_gnn or args.repeat_mask:
            # the reference's encoders read `feature_list` / `mask_x`, which they never define (NameError)
            raise NotImplementedError("dense_gnn / repeat_mask are not usable with the pre-training encoders")
        self.node_num = 5135
        self.pca_prelinear = False
        self.decoder_dim = args.decoder_dim
        self.decoder_type = args.decoder_type
        C, k = args.final_channels, args.pca_dim
        if self.decoder_type == "flatten":
            self.decoder = nn.ModuleList([nn.Linear(C * N_PATHWAYS * k * N_OMICS, self.decoder_dim), nn.ReLU(),
                                          nn.Linear(self.decoder_dim, self.decoder_dim), nn.ReLU(),
                                          nn.Linear(self.decoder_dim, self.node_num * N_OMICS)])
        elif self.decoder_type in ("foreach", "foreach_diffhidden"):
            counts = torch.bincount(torch.as_tensor(pathway_indexs).reshape(-1).long())
            blocks = []
            for n_out in counts.tolist():
                hid = self.decoder_dim if self.decoder_type == "foreach" else next_power_of_two(int(math.sqrt(n_out * C)))
                blocks.append(nn.Sequential(nn.Linear(C * k, hid), nn.ReLU(), nn.Linear(hid, n_out)))
            self.decoder = nn.ModuleList(blocks)
            # output gene g is produced by block _out_block[g] (blocks are concatenated in order)
            self.register_buffer("_out_block", torch.repeat_interleave(torch.arange(len(counts)), counts),
                                 persistent=False)

    def _build_diff_pooling(self, args):
        if args.reorder_type == "diff_pooling":
            feat = {"pathway": args.final_channels, "head": args.conv_channel_list[-1]}.get(args.diff_pooling_location)
            if feat is not None:
                self.diff_pooling = DiffPool(feat, 2, args.pathway_num, args.diff_pooling_layer,
                                             args.diff_pooling_hidden_dim, args.diff_pooling_output_dim, args)

    def _project(self, input_batch, strict_mask=False):
        """-> ``(pooled [B,C,438,k] -- [B,C,146,3k] for the flatten decoder --, gene_feature [B,G,C])``.
        ``strict_mask``: the AutoEncoder's ``match > 0`` instead of ``match >= 0`` (autoencoder.py:106)."""
        args = self.args
        if args.reduction_method != "linear_projection":
            raise NotImplementedError("reduction_method=%r (CPU SVD branch) is outside the accelerated path"
                                      % (args.reduction_method,))
        nodes_per_graph = self.node_num * N_OMICS
        x = input_batch.x.reshape(-1, 1)
        if args.node_embedding:
            x = (x.reshape(-1, nodes_per_graph, 1) * self.node_embedding).reshape(-1, self.node_embedding.shape[-1])
        edge_index, edge_attr = input_batch.edge_index.to(x.device), input_batch.edge_attr.to(x.device)
        for layer in self.gnn_model:
            x = layer(x, edge_index, edge_attr) + x if args.resgnn else layer(x, edge_index, edge_attr)

        match = input_batch.gene_pca_match.to(x.device)
        B, G = match.shape
        idx = match + torch.arange(B, device=x.device)[:, None] * nodes_per_graph
        gene_feature = x[idx]
        if args.pca_match_mask:
            live = (match > 0) if strict_mask else (match >= 0)
            gene_feature = gene_feature * live.to(x.dtype)[:, :, None]
        # projection pooling over the gathered rows themselves (so that a caller's gene_feature.retain_grad() sees
        # the gradient, as get_vae_sim_loss's grad_weight option expects): identity membership, G rows per graph
        ident = torch.arange(G, device=x.device)[None, :].expand(B, G)
        pooled = segment_project(gene_feature.reshape(B * G, -1), ident, input_batch.raw_indice.to(x.device),
                                 self.learnable_pca_params * self.info_mask, G, N_PATHWAYS * N_OMICS,
                                 match_mask=False)                                          # [B, C, 438, k]
        if self.decoder_type == "flatten":
            pooled = pooled.reshape(B, pooled.shape[1], N_PATHWAYS, self.pca_dim * N_OMICS)
        return pooled, gene_feature

    def flatten_decoder(self, h):
        x = h.flatten(1)
        for layer in self.decoder:
            x = layer(x)
        return x

    def foreach_decoder(self, h):
        """``cat_i decoder[i](h[:, i, :])`` -> ``[B, n_genes]``."""
        blocks = list(self.decoder)
        if len({b[0].out_features for b in blocks}) != 1:            # ragged hidden widths: block by block
            return torch.cat([b(h[:, i, :]) for i, b in enumerate(blocks)], dim=-1)
        w1 = torch.stack([b[0].weight for b in blocks])              # [P, D, H]
        b1 = torch.stack([b[0].bias for b in blocks])                # [P, D]
        w2 = torch.cat([b[2].weight for b in blocks], dim=0)         # [n_genes, D]
        b2 = torch.cat([b[2].bias for b in blocks], dim=0)           # [n_genes]
        hid = F.relu(torch.baddbmm(b1[:, None, :], h.permute(1, 0, 2), w1.transpose(1, 2)))     # [P, B, D]
        rows = hid.index_select(0, self._out_block)                  # [n_genes, B, D]: the hidden row each gene reads
        return (rows * w2[:, None, :]).sum(-1).t() + b2


class VAE(_PretrainBase):

    def __init__(self, args, pca_params=None, pathway_indexs=None):
        super().__init__(args, pca_params, pathway_indexs)
        self._build_diff_pooling(args)
        H = args.final_channels * args.pca_dim
        self.enc_mu = nn.Linear(H, H)
        self.enc_log_sigma = nn.Linear(H, H)
        self.init_weight()

    # ------------------------------------------------------------------ encoder / decoders
    def encoder(self, input_batch):
        """-> ``(q_z, cat([mu, sigma], -1), [loss_std, 0, loss_corr], gene_feature [B,G,C])``."""
        pooled, gene_feature = self._project(input_batch)
        x = pooled.permute(0, 2, 1, 3).flatten(2)
        mu = self.enc_mu(x)
        sigma = torch.exp(self.enc_log_sigma(x))
        loss_std = -mu.flatten(1).permute(1, 0).std(1).mean()
        loss_corr = self._mean_abs_offdiag_corr(mu)
        return (torch.distributions.Normal(loc=mu, scale=sigma + 1e-7), torch.cat([mu, sigma], dim=-1),
                [loss_std, 0, loss_corr], gene_feature)

    @staticmethod
    def _mean_abs_offdiag_corr(mu):
        """mean over pathways p and feature pairs (i, j) of |corrcoef(mu[:, p, :].T)[i, j]| with the diagonal zeroed
        (:205-206), as one batched covariance instead of a corrcoef call per pathway."""
        m = mu.permute(1, 2, 0)                                  # [P, H, B]: variables x observations
        m = m - m.mean(dim=2, keepdim=True)
        cov = m @ m.transpose(1, 2) / (m.shape[2] - 1)
        d = torch.sqrt(torch.diagonal(cov, dim1=1, dim2=2))
        corr = (cov / d[:, :, None] / d[:, None, :]).clamp(-1, 1)     # torch.corrcoef clips too
        eye = torch.eye(corr.shape[-1], device=mu.device, dtype=mu.dtype)
        return (corr * (1 - eye)).abs().mean()

    def forward(self, input_batch, x=None, gene_pca_match=None, raw_indice=None, age=None):
        q_z, h, loss, _ = self.encoder(input_batch)
        z = q_z.rsample()
        output = self.flatten_decoder(z) if self.decoder_type == "flatten" else self.foreach_decoder(z)
        return {"pred_x": output, "embedding": h, "q_z": q_z, "z": z, "loss": loss}

    # ------------------------------------------------------------------ prediction path
    def _latent_image(self, h, keep=None):
        """[B, 438, c] -> [B, 1, 146, 3 c'] (``channel_one``) or [B, c', 146, 3]; c' = the mean half (c // 2) of the
        VAE's cat([mu, sigma]) unless ``keep`` says otherwise."""
        b, _, c = h.shape
        keep = c // 2 if keep is None else keep
        if self.args.channel_one:
            h = h[:, :, :keep].reshape(b, 1, N_PATHWAYS, -1)
        else:
            h = h[:, :, :keep].permute(0, 2, 1).reshape(b, keep, N_PATHWAYS, N_OMICS)
        if self.args.reorder_pathway and self.reorder_idxs is not None:
            h = h[:, :, self.reorder_idxs, :]
        return h

    def train_step(self, input_batch, require_grad=True):
        with torch.enable_grad() if require_grad else torch.no_grad():
            q_z, h, _, gene_feature = self.encoder(input_batch)
            if self.args.vae_generate_train_sample:
                h = q_z.rsample()
            # (the reorder below sits outside the grad context in the reference; a pure index_select either way)
            h = self._latent_image(h)
        pred, pca_feature, link, ent = self.predict_head(h, input_batch.age)
        return pred, pca_feature, link, ent, gene_feature

    def eval_step(self, input_batch, require_grad=True):
        with torch.enable_grad() if require_grad else torch.no_grad():
            _, h, _, _ = self.encoder(input_batch)
            h = self._latent_image(h)
        return self.predict_head(h, input_batch.age)

    def predict_head(self, x, age):
        """-> ``(pred [B,2], pca_feature, link_loss, entropy_loss)``; DiffPool either on the latent pathway features
        ('pathway') or behind the conv stack ('head'), otherwise conv -> max-pool (unless 'no_pooling')."""
        args = self.args
        link = ent = 0
        pca_feature = x
        by_diffpool = args.reorder_type == "diff_pooling" and args.diff_pooling_location in ("pathway", "head")
        if not (by_diffpool and args.diff_pooling_location == "pathway"):
            for layer in self.conv_model:
                x = layer(x)
        if by_diffpool:
            b = x.shape[0]
            width = args.final_channels if args.diff_pooling_location == "pathway" else args.conv_channel_list[-1]
            x = x.permute(0, 3, 2, 1).reshape(-1, args.pathway_num, width)
            x, link, ent = self.diff_pooling(x, self.get_pathway_adj().to(x.device))
            x = self.drop1(x.reshape(b, -1))
        else:
            if args.reorder_type != "no_pooling":
                x = self.pooling(x)
            x = torch.flatten(self.drop1(x), start_dim=1)
        if args.use_age:
            x = torch.cat([x, age[:, None]], dim=-1)
        return self.head(x), pca_feature, link, ent

    def reconstruct_head(self, args):
        """Re-create ``self.head`` for the pooling in use (:267-299); DiffPool leaves ceil(146 * 0.25^L) clusters."""
        age = 1 if self.args.use_age else 0
        if args.reorder_type == "no_pooling":
            in_dim = args.conv_channel_list[-1] * N_PATHWAYS * (N_OMICS * self.pca_dim) + age
        elif args.reorder_type == "diff_pooling":
            clusters = self.args.pathway_num
            for _ in range(self.args.diff_pooling_layer):
                clusters = math.ceil(clusters * 0.25)
            in_dim = self.args.diff_pooling_output_dim * clusters * (N_OMICS * self.pca_dim) + age
        else:
            in_dim = args.conv_channel_list[-1] * (N_PATHWAYS // self.pathway_pool_dim) * \
                ((N_OMICS * self.pca_dim) // self.pca_pool_dim) + age
        self.head = _head(in_dim, self.head_dim)
        _xavier(self.head)

    def get_pathway_adj(self):
        if self.args.pathway_similarity == "correlation":
            return self.pathway_similarity_matrix

    def set_pathway_similarity_matrix(self, pathway_similarity_matrix):
        self.pathway_similarity_matrix = (torch.as_tensor(pathway_similarity_matrix) +
                                          torch.eye(self.args.pathway_num)).to(torch.float32)

    def get_embedding_similarity(self):
        raise NotImplementedError("get_embedding_similarity reads spreadsheet embeddings from disk (ETL): outside the "
                                  "accelerated path")

    # ------------------------------------------------------------------ losses
    def vae_loss(self, x_predict, x, z, q_z):
        n = x.size(0)
        recons_loss = F.mse_loss(x_predict, x)
        mmd_loss = torch.stack([self.compute_mmd(z[:, i, :]) for i in range(z.shape[1])]).mean()
        kld_loss = torch.distributions.kl_divergence(q_z, torch.distributions.Normal(0, 1.)).sum(-1).mean()
        a = self.args
        loss = a.mmd_beta * recons_loss + (1. - a.mmd_alpha) * a.kld_weight * kld_loss + \
            (a.mmd_alpha + a.mmd_reg_weight - 1.) / (n * (n - 1)) * mmd_loss
        return {'loss': loss, 'Reconstruction_Loss': recons_loss, 'MMD': mmd_loss, 'KLD': -kld_loss}

    def set_std_weight(self, std_weight):
        self.std_weight = std_weight

    def get_vae_sim_loss(self, pred, ground_y, grad_feat=None):
        a = self.args
        if a.std_weight:
            loss = a.std_weight_coef * (self.std_weight.to(pred.device) * F.l1_loss(pred, ground_y, reduction="none")).mean()
        else:
            loss = F.l1_loss(pred.to(torch.float32), ground_y.to(torch.float32))
        if a.grad_weight and grad_feat is not None:
            w = grad_feat.grad.permute(1, 0, 2).flatten(1).abs().mean(1)
            loss = loss + a.grad_weight_coef * ((w / w.max()) * F.l1_loss(pred, ground_y, reduction="none")).mean()
        return loss

    def compute_kernel(self, x1, x2):
        n, d = x1.size(0), x1.size(1)
        x1 = x1.unsqueeze(-2).expand(n, n, d)
        x2 = x2.unsqueeze(-3).expand(n, n, d)
        if self.args.mmd_kernel_type == 'rbf':
            return self.compute_rbf(x1, x2)
        if self.args.mmd_kernel_type == 'imq':
            return self.compute_inv_mult_quad(x1, x2)
        raise ValueError('Undefined kernel type.')

    def compute_rbf(self, x1, x2, eps=1e-7):
        sigma = 2. * x2.size(-1) * self.args.z_var
        return torch.exp(-((x1 - x2).pow(2).mean(-1) / sigma))

    def compute_inv_mult_quad(self, x1, x2, eps=1e-7):
        c = 2 * x2.size(-1) * self.args.z_var
        kernel = c / (eps + c + (x1 - x2).pow(2).sum(dim=-1))
        return kernel.sum() - kernel.diag().sum()           # off-diagonal mass

    def compute_mmd(self, z):
        z = z.reshape(-1, z.shape[-1])
        prior = torch.randn_like(z)
        return self.compute_kernel(prior, prior).mean() + self.compute_kernel(z, z).mean() - \
            2 * self.compute_kernel(prior, z).mean()


class VectorQuantizer(nn.Module):
    """Nearest-code-word quantiser with commitment + embedding loss and a straight-through gradient (reference
    vq_vae.py:36-82, after the sonnet VQ-VAE).  The code vector is gathered by index instead of multiplying a one-hot
    matrix by the codebook."""

    def __init__(self, num_embeddings, embedding_dim, beta=0.25):
        super().__init__()
        self.K, self.D, self.beta = num_embeddings, embedding_dim, beta
        self.embedding = nn.Embedding(self.K, self.D)
        self.embedding.weight.data.uniform_(-1, 1)

    def forward(self, latents):
        flat = latents.reshape(-1, self.D)
        w = self.embedding.weight
        dist = (flat ** 2).sum(1, keepdim=True) + (w ** 2).sum(1) - 2 * flat @ w.t()
        q = self.embedding(torch.argmin(dist, dim=1)).view(latents.shape)
        vq_loss = F.mse_loss(q.detach(), latents) * self.beta + F.mse_loss(q, latents.detach())
        return latents + (q - latents).detach(), vq_loss


class VQ_VAE(VAE):
    """Reference ``models/vq_vae.py``: the VAE's surface with a vector-quantised latent -- ``encoder`` returns the
    pooled latent itself, ``forward`` quantises it before the decoders, ``train_step`` / ``eval_step`` feed the
    un-quantised latent to ``predict_head``.  (``enc_mu`` / ``enc_log_sigma`` exist, unused, as in the reference.)"""

    def __init__(self, args, pca_params=None, pathway_indexs=None):
        super().__init__(args, pca_params, pathway_indexs)
        if args.vae_generate_train_sample:
            raise NotImplementedError("vae_generate_train_sample reads an undefined q_z in the reference's VQ_VAE")
        self.vq_layer = VectorQuantizer(args.vqvae_num_embeddings, args.final_channels * args.pca_dim, args.vqvae_beta)
        self.init_weight()

    def encoder(self, input_batch):
        pooled, _ = self._project(input_batch)
        return pooled.permute(0, 2, 1, 3).contiguous().flatten(2)

    def train_step(self, input_batch, require_grad=True):
        with torch.enable_grad() if require_grad else torch.no_grad():
            h = self.encoder(input_batch)
            h = self._latent_image(h, keep=h.shape[-1])
        return self.predict_head(h, input_batch.age)

    eval_step = train_step

    def forward(self, input_batch, x=None, gene_pca_match=None, raw_indice=None, age=None):
        z = self.encoder(input_batch)
        quantized_z, vq_loss = self.vq_layer(z)
        output = self.flatten_decoder(quantized_z) if self.decoder_type == "flatten" else self.foreach_decoder(quantized_z)
        return {"pred_x": output, "embedding": quantized_z, "z": z, "vq_loss": vq_loss}

    def vae_loss(self, x_predict, x, vq_loss):
        recons_loss = F.mse_loss(x_predict, x)
        return {'loss': self.args.mmd_beta * recons_loss + vq_loss, 'Reconstruction_Loss': recons_loss,
                'vq_loss': vq_loss}


class AutoEncoder(_PretrainBase):
    """Reference ``models/autoencoder.py``: ``forward(batch) -> (reconstruction [B, n_genes], latent, None)``; the
    encoder masks ``match <= 0`` (:106) and hands the 4-D pooled tensor to the decoders."""

    def __init__(self, args, pca_params=None, pathway_indexs=None):
        super().__init__(args, pca_params, pathway_indexs)
        if not args.mutual_info_mask and args.final_channels == 1:
            # the reference's un-masked single-channel branch (:122) builds a 3-D tensor and fails at :124
            raise NotImplementedError("AutoEncoder without mutual_info_mask needs final_channels != 1")
        self.init_weight()

    def encoder(self, input_batch):
        return self._project(input_batch, strict_mask=True)[0]

    def foreach_decoder(self, h):
        return super().foreach_decoder(h.permute(0, 2, 1, 3).flatten(2))

    def forward(self, input_batch, x=None, gene_pca_match=None, raw_indice=None, age=None):
        h = self.encoder(input_batch)
        output = self.flatten_decoder(h) if self.decoder_type == "flatten" else self.foreach_decoder(h)
        return output, h, None
