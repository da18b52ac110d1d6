"""Reference model forwards, restated as pure functions of ``(args, state_dict, batch)``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``args`` is the reference's
``opt.py`` namespace (any object with the same attribute names), ``sd`` a
``state_dict`` with the reference's key names, ``batch`` a duck-typed batch
(SURVEY.md section 8b).  Only the branches the hot-path scope table (section
8a) lists are restated; anything else raises ``NotImplementedError``.
"""
from math import ceil

import torch
import torch.nn.functional as F

from . import gcn_lib as G
from . import primitives as P


def _dropout(x, p, training):
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


# ----------------------------------------------------------------------------
# models/deepergcn.py
# ----------------------------------------------------------------------------
def deepergcn_forward(args, sd, batch, training=False):
    """``DeeperGCN.forward`` (deepergcn.py:185-323) for ``gnn_encoder='linear'``,
    ``conv='gen'``; blocks res+/res/plain (:232-281); readout by
    ``global_*_pool`` (:319) or the pathway-global-node branch with
    ``pathway_readout in (None, 'maxpool')`` (:283-317).  Dropout is applied
    only when ``training`` and ``args.dropout > 0`` (parity runs use 0)."""
    if args.gnn_encoder != "linear" or getattr(args, "pca_only", False):
        raise NotImplementedError("only gnn_encoder='linear', pca_only=False is on the hot path")
    x, ei = batch.x, batch.edge_index
    ea = batch.edge_attr.to(torch.long) if args.global_edge == "onehot" else batch.edge_attr
    L = args.num_layers

    if args.node_embedding:
        emb = F.embedding(x[:, -1].to(torch.long), sd["node_embedding_encoder.weight"])
        h = F.linear(torch.cat([x[:, :-1], emb], dim=-1),
                     sd["node_features_encoder.weight"], sd["node_features_encoder.bias"])
    else:
        h = F.linear(x, sd["node_features_encoder.weight"], sd["node_features_encoder.bias"])

    if args.use_edge_attr:
        if args.global_edge == "onehot":
            edge_emb = F.embedding(ea, sd["edge_encoder.weight"])      # [E,1,H]; GENConv flattens
        else:
            edge_emb = F.linear(ea, sd["edge_encoder.weight"], sd["edge_encoder.bias"])
    else:
        edge_emb = None

    ends = None
    if args.pathway_global_node:
        pn = args.pathway_num
        pemb = F.linear(batch.pathway_node_attr, sd["pathway_features_encoder.weight"],
                        sd["pathway_features_encoder.bias"])
        pemb = pemb.reshape(-1, pemb.shape[-1])
        ends = torch.cumsum(batch.node_size, dim=0).tolist()
        h = h.clone()
        for i, end in enumerate(ends):
            h[end - pn:end] = pemb[i * pn:(i + 1) * pn]

    def conv(l, inp):
        return G.genconv(inp, ei, edge_emb, sd, "gcns.%d." % l, aggr=args.gcn_aggr, t=args.t,
                         learn_t=args.learn_t, p=args.p, learn_p=args.learn_p,
                         msg_norm_on=args.msg_norm, encode_edge=args.conv_encode_edge,
                         norm_kind=args.norm, mlp_layers=args.mlp_layers, training=training)

    def nrm(l, inp):
        return G.norm(inp, args.norm, sd, "norms.%d." % l, training)

    pd = args.dropout
    if args.block == "res+":
        h = conv(0, h)
        for l in range(1, L):
            h1 = h if args.no_inter_norm else nrm(l - 1, h)
            h2 = F.relu(h1)
            if not args.no_inter_drop:
                h2 = _dropout(h2, pd, training)
            h = conv(l, h2) + h
        h = nrm(L - 1, h)
        if not args.no_inter_drop:
            h = _dropout(h, pd, training)
    elif args.block == "res":
        h = _dropout(F.relu(nrm(0, conv(0, h))), pd, training)
        for l in range(1, L):
            h = F.relu(nrm(l, conv(l, h))) + h
            h = _dropout(h, pd, training)
    elif args.block == "plain":
        h = _dropout(F.relu(nrm(0, conv(0, h))), pd, training)
        for l in range(1, L):
            h1 = conv(l, h)
            h2 = h1 if args.no_inter_norm else nrm(l, h1)
            h = F.relu(h2) if l != L - 1 else h2
            if not args.no_inter_drop:
                h = _dropout(h, pd, training)
    else:
        raise NotImplementedError(args.block)

    if args.pathway_global_node:
        pn = args.pathway_num
        rows = torch.stack([h[end - pn:end] for end in ends])          # [B, pn, H]
        if args.pathway_readout is None:
            pb = torch.cat([batch.batch[end - pn:end] for end in ends])
            h_graph = P.global_pool(rows.reshape(-1, rows.shape[-1]), pb, args.graph_pooling)
        elif args.pathway_readout == "maxpool":
            if args.feature_drop:
                rows = _dropout(rows, 0.25, training)
            h_graph = torch.flatten(F.max_pool1d(rows.transpose(1, 2), 4), start_dim=1)
            if args.pre_concat_age:
                h_graph = torch.cat([h_graph, batch.age[:, None]], dim=-1)
            h_graph = F.relu(F.linear(h_graph, sd["readout_func.0.weight"], sd["readout_func.0.bias"]))
            if not args.pre_readout_drop:
                h_graph = _dropout(h_graph, 0.5, training)
        else:
            raise NotImplementedError(args.pathway_readout)
    else:
        h_graph = P.global_pool(h, batch.batch, args.graph_pooling)

    if args.use_age and not args.pre_concat_age:
        h_graph = torch.cat([h_graph, batch.age[:, None]], dim=-1)
    z = h_graph
    for i in range(args.num_layer_head - 1):
        z = F.relu(F.linear(z, sd["graph_pred_linear.%d.weight" % (2 * i)],
                            sd["graph_pred_linear.%d.bias" % (2 * i)]))
        if args.head_dropout:
            z = _dropout(z, pd, training)
    k = 2 * args.num_layer_head
    z = F.linear(z, sd["graph_pred_linear.%d.weight" % k], sd["graph_pred_linear.%d.bias" % k])
    return F.softmax(z, dim=-1)


# ----------------------------------------------------------------------------
# models/multilevel_gnn.py
# ----------------------------------------------------------------------------
def projection_pool(x_nodes, gene_pca_match, raw_indice, pca_params, info_mask, nodes_per_graph,
                    n_segments, match_mask=True):
    """Gene -> pathway learnable-projection pooling (multilevel_gnn.py:212-239), literal form.

    ``x_nodes [B*NN, C]``; ``gene_pca_match [B, G]`` (node index inside the
    graph, ``-1`` = absent; NOTE a negative index wraps like the reference's
    advanced indexing does); ``raw_indice [B, G]`` segment id; returns
    ``[B, C, n_segments, k]`` (before the reshape to ``[B, C, 146, 3k]``).
    """
    B, Gn = gene_pca_match.shape
    C = x_nodes.shape[1]
    k = pca_params.shape[1]
    idx = gene_pca_match + torch.arange(B, device=x_nodes.device)[:, None] * nodes_per_graph
    xg = x_nodes[idx]                                                   # [B, G, C]
    if match_mask:
        xg = xg * torch.where(gene_pca_match >= 0, 1, 0)[:, :, None]
    weights = pca_params * info_mask if info_mask is not None else pca_params
    res = xg.unsqueeze(3).repeat(1, 1, 1, k) * weights[:, None, :]      # [B, G, C, k]
    res = res.permute(0, 2, 1, 3)                                       # [B, C, G, k]
    ridx = raw_indice[:, None, :, None].repeat(1, C, 1, k)
    out = torch.zeros(B, C, n_segments, k, dtype=x_nodes.dtype, device=x_nodes.device)
    return out.scatter_reduce(2, ridx, res, reduce="sum")


def multilevel_gnn_forward(args, sd, batch, node_num, training=False, reorder_idxs=None,
                           n_pathways=146, head_prefix="", only_mrna_pred=False):
    """``MultilevelGNN.forward`` (multilevel_gnn.py:132-292), default wiring:
    ``reduction_method='linear_projection'``, single edge tensor, ``used_omics='012'``,
    no ``pca_compare``/``pca_prelinear``.  Returns ``(pred, pca_feature)``.
    ``head_prefix`` / ``only_mrna_pred``: the ``PathwayHeadSeq`` form of the head (see
    :func:`multilevel_gnn_seq_forward`)."""
    if args.reduction_method != "linear_projection" or args.pca_compare or args.pca_prelinear:
        raise NotImplementedError
    NN = node_num * 3
    mask_x = batch.x
    x = batch.x.reshape(-1, 1)
    if args.node_embedding:
        emb = sd["node_embedding"]
        x = (x.reshape(-1, NN, 1) * emb).reshape(-1, emb.shape[-1])
    ei = batch.edge_index
    ea = batch.edge_attr if args.weighted_edge else None
    n_layers = len([k for k in sd if k.startswith("gnn_model.") and k.endswith("gconv.lin_r.weight")])
    feats = []
    for i in range(n_layers):
        last = i == n_layers - 1
        y = G.sageconv(x, ei, ea, sd, "gnn_model.%d.gconv." % i, act_name=args.gnn_act,
                       relative=(args.gnn_name.lower() == "rsage"),
                       normalize=bool(args.gnn_last_norm) if last else False,
                       mlp_norm=args.gnn_mlp_norm, training=training)
        if args.dense_gnn:
            x = y
            feats.append(x)
        elif args.resgnn:
            x = y + x
        else:
            x = y
        if not last and args.repeat_mask and (i + 1) % args.repeat_cyclic == 0:
            if args.repeat_norm:
                x = x / (x ** 2).sum(1).sqrt()[:, None]
            x = x * mask_x.reshape(-1, 1)
    if args.dense_gnn:
        x = torch.cat(feats, dim=-1)
    if args.value_att_mask:
        if args.merge_mode == "mult":
            x = x * mask_x.reshape(-1, 1)
        else:
            x = args.add_coef1 * x + args.add_coef2 * mask_x.reshape(-1, 1)
    k = sd["learnable_pca_params"].shape[1]
    n_omics = 3
    pooled = projection_pool(x, batch.gene_pca_match, batch.raw_indice, sd["learnable_pca_params"],
                             sd.get("info_mask"), NN, n_pathways * n_omics, args.pca_match_mask)
    B, C = pooled.shape[0], pooled.shape[1]
    x = pooled.reshape(-1, C, n_pathways, k * n_omics)
    if args.reorder_pathway and reorder_idxs is not None:
        x = x[:, :, reorder_idxs, :]
    pca_feature = x

    hp = head_prefix
    n_conv = len([kk for kk in sd if kk.startswith(hp + "conv_model.") and kk.endswith(".weight")])
    for i in range(n_conv):
        w = sd[hp + "conv_model.%d.weight" % (2 * i)]
        x = F.relu(F.conv2d(x, w, sd[hp + "conv_model.%d.bias" % (2 * i)], padding=w.shape[-1] // 2))
    if only_mrna_pred:                       # PathwayHeadSeq.forward :61-64: first two columns, no dropout
        x = F.max_pool2d(x[:, :, :, :2], (args.pathway_pool_dim, args.pca_pool_dim))
    else:
        x = F.max_pool2d(x, (args.pathway_pool_dim, args.pca_pool_dim))
        x = _dropout(x, 0.25 if args.feature_drop else 0.0, training)
    x = torch.flatten(x, start_dim=1)
    if args.use_age:
        x = torch.cat([x, batch.age[:, None]], dim=-1)
    x = F.relu(F.linear(x, sd[hp + "head.0.weight"], sd[hp + "head.0.bias"]))
    x = _dropout(x, 0.5, training)
    x = F.linear(x, sd[hp + "head.3.weight"], sd[hp + "head.3.bias"])
    return F.softmax(x, dim=1), pca_feature      # nn.Softmax() implicit dim=1 for 2-D input


def multilevel_gnn_seq_forward(args, sd, batch, node_num, training=False, reorder_idxs=None, n_pathways=146):
    """``MultilevelGNNSeq.forward`` (multilevel_gnn_seq.py:157-291): the body of ``MultilevelGNN.forward`` up to
    ``pca_feature``, then ``PathwayHeadSeq`` (:14-68) -- the conv stack, max-pool, ``drop1``, flatten, age, MLP head
    held by a submodule (``state_dict`` keys ``pathwayhead.conv_model.*`` / ``pathwayhead.head.*``); with
    ``only_mrna_pred`` the conv output is cut to its first two columns before pooling and ``drop1`` is skipped."""
    return multilevel_gnn_forward(args, sd, batch, node_num, training, reorder_idxs, n_pathways,
                                  head_prefix="pathwayhead.", only_mrna_pred=bool(args.only_mrna_pred))


def feature_loss(args, sd, pca_feature, pathway_indexs=None):
    """``MultilevelGNN.get_feature_loss`` (multilevel_gnn.py:329-348).  The independence
    term is computed on ``.data`` in the reference: it changes the value, never the gradient;
    its accumulation sits outside the inner loop (:345)."""
    loss = 0
    if args.pca_loss:
        flat = pca_feature.reshape(pca_feature.shape[0], -1)
        loss = loss - args.pca_loss_coef * torch.log(torch.mean(torch.std(flat, dim=0)))
    if args.pca_indep_loss:
        w = (sd["learnable_pca_params"] * sd["info_mask"]).detach()
        k = w.shape[1]
        n = int(pathway_indexs.max().item()) + 1
        indep, count = 0, 0
        for i in range(k - 1):
            for j in range(i + 1, k):
                count += 1
                z = torch.zeros(n, dtype=w.dtype)
                mul = z.scatter_reduce(0, pathway_indexs, w[:, i] * w[:, j], reduce="sum")
                li = z.scatter_reduce(0, pathway_indexs, w[:, i] ** 2, reduce="sum")
                lj = z.scatter_reduce(0, pathway_indexs, w[:, j] ** 2, reduce="sum")
                ln = torch.sqrt(li * lj)
            indep = indep + torch.mean(torch.abs(mul / (ln + 1e-7)))
        loss = loss + indep / count
    return loss


# ----------------------------------------------------------------------------
# models/diff_pooling.py
# ----------------------------------------------------------------------------
def _sage_convolutions(x, adj, sd, prefix, num_layers, training=False):
    """``SAGEConvolutions.forward`` (diff_pooling.py:34-46)."""
    for i in range(num_layers - 1):
        lp = prefix + "layers.%d." % i
        xn = F.relu(P.dense_sage_conv(x, adj, sd[lp + "lin_rel.weight"], sd[lp + "lin_root.weight"],
                                      sd[lp + "lin_root.bias"], True))
        b, n, c = xn.shape
        bp = prefix + "bns.%d." % i
        xn = F.batch_norm(xn.reshape(-1, c), None if training else sd[bp + "running_mean"],
                          None if training else sd[bp + "running_var"], sd[bp + "weight"], sd[bp + "bias"],
                          training, 0.1, 1e-5).view(b, n, c)
        x = x + xn if x.shape == xn.shape else xn
    lp = prefix + "layers.%d." % (num_layers - 1)
    return P.dense_sage_conv(x, adj, sd[lp + "lin_rel.weight"], sd[lp + "lin_root.weight"],
                             sd[lp + "lin_root.bias"], True)


def diffpool_forward(sd, x, adj, num_layers, after_pooling_layer=1, prefix="", training=False):
    """``DiffPool.forward`` (diff_pooling.py:116-133) with ``DiffPoolLayer`` (:60-65).
    Returns ``(x, link_total, ent_total)``."""
    l_total, e_total = 0, 0
    for i in range(num_layers):
        lp = prefix + "diffpool_layers.%d." % i
        s = _sage_convolutions(x, adj, sd, lp + "gnn_pool.", 1, training)
        z = _sage_convolutions(x, adj, sd, lp + "gnn_embed.", 1, training)
        x, adj, link, ent = P.dense_diff_pool(z, adj, s)
        x = _sage_convolutions(x, adj, sd, prefix + "after_pool_layers.%d." % i, after_pooling_layer,
                               training)
        l_total = l_total + link
        e_total = e_total + ent
    return x, l_total, e_total


def diffpool_cluster_sizes(max_num_nodes, num_layers):
    """Cluster counts per level (diff_pooling.py:85,92,106): ``ceil`` of 0.25 (0.1 if one layer)."""
    f = 0.1 if num_layers == 1 else 0.25
    sizes, n = [], max_num_nodes
    for _ in range(num_layers):
        n = ceil(f * n)
        sizes.append(n)
    return sizes


# ----------------------------------------------------------------------------
# models/vae.py  (VAE(MultilevelGNN): encoder, decoders, predict_head with DiffPool)
# ----------------------------------------------------------------------------
def _latent_projection(args, sd, batch, node_num, decoder_type="foreach", n_pathways=146, strict_mask=False):
    """Shared front of the three pre-training models' ``encoder`` (vae.py:128-183, vq_vae.py:177-229,
    autoencoder.py:70-127), ``reduction_method='linear_projection'``: GraphConv stack (edge attributes always passed,
    no value mask), gather by ``gene_pca_match`` (masked where ``match < 0``; the AutoEncoder masks ``match <= 0``,
    ``strict_mask``), projection with ``learnable_pca_params * info_mask``, segment sum.  ``dense_gnn`` /
    ``repeat_mask`` read names the reference never defines there (NameError) and are rejected.
    Returns ``(pooled [B,C,438,k] or [B,C,146,3k] for the flatten decoder, gene_feature [B,G,C])``."""
    if args.reduction_method != "linear_projection" or args.dense_gnn or args.repeat_mask:
        raise NotImplementedError
    NN = node_num * 3
    x = batch.x.reshape(-1, 1)
    if args.node_embedding:
        emb = sd["node_embedding"]
        x = (x.reshape(-1, NN, 1) * emb).reshape(-1, emb.shape[-1])
    n_layers = len([k for k in sd if k.startswith("gnn_model.") and k.endswith("gconv.lin_r.weight")])
    for i in range(n_layers):
        last = i == n_layers - 1
        y = G.sageconv(x, batch.edge_index, batch.edge_attr, sd, "gnn_model.%d.gconv." % i, act_name=args.gnn_act,
                       relative=(args.gnn_name.lower() == "rsage"),
                       normalize=bool(args.gnn_last_norm) if last else False,
                       mlp_norm=args.gnn_mlp_norm, training=False)
        x = y + x if args.resgnn else y
    match = batch.gene_pca_match
    B = match.shape[0]
    idx = match + torch.arange(B)[:, None] * NN
    gene_feature = x[idx]
    if args.pca_match_mask:
        live = (match > 0) if strict_mask else (match >= 0)
        gene_feature = gene_feature * torch.where(live, 1, 0)[:, :, None]
    k = sd["learnable_pca_params"].shape[1]
    C = gene_feature.shape[-1]
    w = sd["learnable_pca_params"] * sd["info_mask"]
    res = gene_feature.unsqueeze(3).repeat(1, 1, 1, k) * w[:, None, :]                 # [B,G,C,k]
    res = res.permute(0, 2, 1, 3)                                                      # [B,C,G,k]
    ridx = batch.raw_indice[:, None, :, None].repeat(1, C, 1, k)
    pooled = torch.zeros(B, C, n_pathways * 3, k, dtype=res.dtype).scatter_reduce(2, ridx, res, reduce="sum")
    if decoder_type == "flatten":
        pooled = pooled.reshape(-1, C, n_pathways, k * 3)
    return pooled, gene_feature


def vae_encoder(args, sd, batch, node_num, decoder_type="foreach", n_pathways=146):
    """``VAE.encoder`` (vae.py:128-208).  Returns ``(mu, sigma, [loss_std, 0, loss_corr], gene_feature)``;
    ``q_z = Normal(mu, sigma + 1e-7)``."""
    pooled, gene_feature = _latent_projection(args, sd, batch, node_num, decoder_type, n_pathways)
    x = pooled.permute(0, 2, 1, 3).flatten(2)
    mu = F.linear(x, sd["enc_mu.weight"], sd["enc_mu.bias"])
    sigma = torch.exp(F.linear(x, sd["enc_log_sigma.weight"], sd["enc_log_sigma.bias"]))
    loss_std = -mu.flatten(1).permute(1, 0).std(1).mean()
    off_diag = torch.ones(mu.shape[-1], mu.shape[-1]) - torch.eye(mu.shape[-1])
    loss_corr = torch.stack([(torch.corrcoef(m) * off_diag).abs() for m in mu.permute(1, 2, 0)]).mean()
    return mu, sigma, [loss_std, 0, loss_corr], gene_feature


def vae_foreach_decoder(sd, z):
    """``VAE.foreach_decoder`` (vae.py:216-222): block i = Linear -> ReLU -> Linear on ``z[:, i, :]``, concatenated."""
    outs = []
    for i in range(z.shape[1]):
        hdn = F.relu(F.linear(z[:, i, :], sd["decoder.%d.0.weight" % i], sd["decoder.%d.0.bias" % i]))
        outs.append(F.linear(hdn, sd["decoder.%d.2.weight" % i], sd["decoder.%d.2.bias" % i]))
    return torch.cat(outs, dim=-1)


def vae_predict_head(args, sd, x, age, adj=None, training=False):
    """``VAE.predict_head`` (vae.py:233-265) -> ``(pred, pca_feature, link, ent)``; ``adj`` = the matrix stored by
    ``set_pathway_similarity_matrix`` (similarity + I, :305-306)."""
    link = ent = 0
    pca_feature = x

    def convs(v):
        n_conv = len([kk for kk in sd if kk.startswith("conv_model.") and kk.endswith(".weight")])
        for i in range(n_conv):
            w = sd["conv_model.%d.weight" % (2 * i)]
            v = F.relu(F.conv2d(v, w, sd["conv_model.%d.bias" % (2 * i)], padding=w.shape[-1] // 2))
        return v

    pooled_by_diffpool = args.reorder_type == "diff_pooling"
    if pooled_by_diffpool and args.diff_pooling_location == "pathway":
        b = x.shape[0]
        x = x.permute(0, 3, 2, 1).reshape(-1, args.pathway_num, args.final_channels)
    elif pooled_by_diffpool and args.diff_pooling_location == "head":
        x = convs(x)
        b = x.shape[0]
        x = x.permute(0, 3, 2, 1).reshape(-1, args.pathway_num, args.conv_channel_list[-1])
    else:
        pooled_by_diffpool = False
        x = convs(x)
        if args.reorder_type != "no_pooling":
            x = F.max_pool2d(x, (args.pathway_pool_dim, args.pca_pool_dim))
        x = _dropout(x, 0.25 if args.feature_drop else 0.0, training)
        x = torch.flatten(x, start_dim=1)
    if pooled_by_diffpool:
        x, link, ent = diffpool_forward(sd, x, adj, args.diff_pooling_layer, args.after_pooling_layer,
                                        prefix="diff_pooling.", training=training)
        x = _dropout(x.reshape(b, -1), 0.25 if args.feature_drop else 0.0, training)
    if args.use_age:
        x = torch.cat([x, age[:, None]], dim=-1)
    x = F.relu(F.linear(x, sd["head.0.weight"], sd["head.0.bias"]))
    x = _dropout(x, 0.5, training)
    x = F.linear(x, sd["head.3.weight"], sd["head.3.bias"])
    return F.softmax(x, dim=1), pca_feature, link, ent


def vae_flatten_decoder(sd, h):
    """``flatten_decoder`` (vae.py:210-214): Linear, ReLU, Linear, ReLU, Linear on the flattened latent."""
    x = h.flatten(1)
    x = F.relu(F.linear(x, sd["decoder.0.weight"], sd["decoder.0.bias"]))
    x = F.relu(F.linear(x, sd["decoder.2.weight"], sd["decoder.2.bias"]))
    return F.linear(x, sd["decoder.4.weight"], sd["decoder.4.bias"])


def vector_quantize(latents, codebook, beta):
    """``VectorQuantizer.forward`` (vq_vae.py:53-82): nearest code word per latent row (squared L2), commitment +
    embedding loss, straight-through estimator.  Returns ``(quantized, vq_loss)``."""
    flat = latents.reshape(-1, codebook.shape[1])
    dist = (flat ** 2).sum(1, keepdim=True) + (codebook ** 2).sum(1) - 2 * flat @ codebook.t()
    q = codebook[torch.argmin(dist, dim=1)].view(latents.shape)
    vq_loss = F.mse_loss(q.detach(), latents) * beta + F.mse_loss(q, latents.detach())
    return latents + (q - latents).detach(), vq_loss


def vq_vae_encoder(args, sd, batch, node_num, decoder_type="foreach"):
    """``VQ_VAE.encoder`` (vq_vae.py:177-231) -> ``[B, 438, C k]`` (``[B, 146, 3 C k]`` for the flatten decoder)."""
    pooled, _ = _latent_projection(args, sd, batch, node_num, decoder_type)
    return pooled.permute(0, 2, 1, 3).contiguous().flatten(2)


def vq_vae_train_step(args, sd, batch, node_num, adj=None, reorder_idxs=None):
    """``VQ_VAE.train_step`` / ``eval_step`` (vq_vae.py:139-165): the un-quantised latent through ``predict_head``."""
    h = vq_vae_encoder(args, sd, batch, node_num)
    b, _, c = h.shape
    h = h.reshape(b, 1, 146, -1) if args.channel_one else h.permute(0, 2, 1).reshape(b, c, 146, 3)
    if args.reorder_pathway and reorder_idxs is not None:
        h = h[:, :, reorder_idxs, :]
    return vae_predict_head(args, sd, h, batch.age, adj)


def vq_vae_forward(args, sd, batch, node_num):
    """``VQ_VAE.forward`` (vq_vae.py:168-175) with the per-pathway decoders -> ``(pred_x, quantized_z, z, vq_loss)``."""
    z = vq_vae_encoder(args, sd, batch, node_num)
    qz, vq_loss = vector_quantize(z, sd["vq_layer.embedding.weight"], args.vqvae_beta)
    return vae_foreach_decoder(sd, qz), qz, z, vq_loss


def autoencoder_forward(args, sd, batch, node_num, decoder_type="foreach"):
    """``AutoEncoder.forward`` (autoencoder.py:62-68, encoder :70-127, decoders :130-142) -> ``(pred_x, h)``; the
    encoder masks ``match <= 0`` (:106) and returns the 4-D pooled tensor, which the decoders flatten themselves."""
    if not args.mutual_info_mask and args.final_channels == 1:
        raise NotImplementedError        # the reference's un-masked single-channel branch builds a 3-D tensor and fails
    h, _ = _latent_projection(args, sd, batch, node_num, decoder_type, strict_mask=True)
    if decoder_type == "flatten":
        return vae_flatten_decoder(sd, h), h
    return vae_foreach_decoder(sd, h.permute(0, 2, 1, 3).flatten(2)), h


def vae_train_step(args, sd, batch, node_num, adj=None, reorder_idxs=None, decoder_type="foreach"):
    """``VAE.train_step`` / ``eval_step`` without the random ``rsample`` option (vae.py:90-117): the mean half of the
    encoder output, laid out as ``[B,1,146,3H]`` (``channel_one``) or ``[B,H,146,3]``, through ``predict_head``."""
    mu, sigma, _, gene_feature = vae_encoder(args, sd, batch, node_num, decoder_type)
    b = mu.shape[0]
    if args.channel_one:
        h = mu.reshape(b, 1, 146, -1)
    else:
        h = mu.permute(0, 2, 1).reshape(b, mu.shape[-1], 146, 3)
    if args.reorder_pathway and reorder_idxs is not None:
        h = h[:, :, reorder_idxs, :]
    pred, feat, link, ent = vae_predict_head(args, sd, h, batch.age, adj)
    return pred, feat, link, ent, gene_feature
