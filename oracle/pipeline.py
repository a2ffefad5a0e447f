"""CPU ORACLE (test infrastructure) -- fp32 torch-CPU restatement of the GarmentNets inference hot path.

Functional: every stage takes the flat checkpoint ``state_dict`` (reference key schema) and
the hyper-parameter dict, so nothing here depends on the product package.  The dense arithmetic uses the
same ATen CPU ops the reference dispatches to (linear, batch_norm, group_norm, conv3d, max_pool3d,
interpolate, grid_sample); the third-party point ops come from ``oracle/gn_oracle.c``.

Reference call sites restated (file:line in /root/reference):
  MLP / PointBatchNorm1D            components/mlp.py:3-20
  SAModule / GlobalSAModule / FP    components/pointnet2.py:11-76 (+ PyG PointConv 1.7.2 semantics)
  PointNet2NOCS.forward             networks/pointnet2_nocs.py:134-166
  pointnet2_forward post-processing networks/conv_implicit_wnf.py:213-240
  VolumeFeatureAggregator.forward   networks/conv_implicit_wnf.py:43-100, components/gridding.py:161-256
  Abstract3DUNet.forward            components/unet3d.py:19-144,195-330,449-474
  ImplicitWNFDecoder.forward        networks/conv_implicit_wnf.py:128-149
  predict loop                      predict.py:138-209
"""
import numpy as np
import torch
import torch.nn.functional as F

import oracle as O

DEFAULT_HPARAMS = {
    # config/train_pointnet2_default.yaml:30-48
    "pointnet2_params": dict(feature_dim=128, batch_norm=True, dropout=True, sa1_ratio=0.5, sa1_r=0.05,
                             sa2_ratio=0.25, sa2_r=0.1, fp3_k=1, fp2_k=3, fp1_k=3, symmetry_axis=None, nocs_bins=64),
    # config/train_pipeline_default.yaml:39-74
    "volume_agg_params": dict(nn_channels=[137, 137, 128], batch_norm=True, lower_corner=[0, 0, 0],
                              upper_corner=[1, 1, 1], grid_shape=[32, 32, 32], reduce_method="max",
                              include_point_feature=True, include_confidence_feature=True),
    "unet3d_params": dict(in_channels=128, out_channels=128, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4),
    "volume_decoder_params": dict(nn_channels=[128, 256, 256, 1], batch_norm=True),
    "surface_decoder_params": dict(nn_channels=[128, 256, 256, 3], batch_norm=True),
    "mc_surface_decoder_params": dict(nn_channels=[128, 256, 256, 1], batch_norm=True),
    "mc_surface_loss_weight": 0,
}


# ------------------------------------------------------------------------------------------------ MLP
def mlp(sd, prefix, x):
    """components/mlp.py:9-20: per layer BN(ReLU(Linear(x))) -- BN after the ReLU, also on the last layer."""
    i = 0
    while f"{prefix}.{i}.0.weight" in sd:
        p = f"{prefix}.{i}"
        x = F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"])
        x = F.relu(x)
        if p + ".2.running_mean" in sd:
            shp = x.shape
            x = F.batch_norm(x.reshape(-1, shp[-1]), sd[p + ".2.running_mean"], sd[p + ".2.running_var"],
                             sd[p + ".2.weight"], sd[p + ".2.bias"], training=False, eps=1e-5).reshape(shp)
        i += 1
    return x


# ------------------------------------------------------------------------------------------------ PointNet++
def sa_module(sd, prefix, x, pos, ptr, ratio, r, self_loops=True, return_graph=False, self_loop_scope="batch"):
    """components/pointnet2.py:22-33 + PyG PointConv(add_self_loops=True, aggr='max').  self_loop_scope "example": the self-loop rule
    applied per example (node i of a centre = point i of its own cloud), i.e. the batch-of-one result for every example"""
    idx, cptr = O.fps(pos.numpy(), ptr, ratio)
    nbr, cnt = O.ball_query(pos.numpy(), ptr, idx, cptr, r, 64)
    M = len(idx)
    nbr_t = torch.from_numpy(nbr.astype(np.int64))
    if self_loops:
        # PointConv quirk on the bipartite graph: remove edges whose numeric source == target index, then
        # add (i, i) for i < M: centre i also receives POINT i of the full cloud.
        ar = torch.arange(M).unsqueeze(1)
        if self_loop_scope == "example":
            ar = torch.cat([ptr[b] + torch.arange(cptr[b + 1] - cptr[b]) for b in range(len(ptr) - 1)]).unsqueeze(1)
        nbr_t = torch.where(nbr_t == ar, torch.full_like(nbr_t, -1), nbr_t)
        nbr_t = torch.cat([nbr_t, ar], dim=1)
    valid = nbr_t >= 0
    tgt = torch.arange(M).unsqueeze(1).expand_as(nbr_t)[valid]
    src = nbr_t[valid]
    cpos = pos[torch.from_numpy(idx)]
    rel = pos[src] - cpos[tgt]
    msg = torch.cat([x[src], rel], dim=1) if x is not None else rel
    msg = mlp(sd, prefix + ".conv.local_nn", msg)
    out = torch.zeros(M, msg.shape[1]).scatter_reduce(0, tgt.unsqueeze(1).expand_as(msg), msg, "amax", include_self=False)
    if return_graph:
        return out, cpos, cptr, dict(idx=idx, nbr=nbr, cnt=cnt)
    return out, cpos, cptr


def global_sa_module(sd, prefix, x, pos, ptr):
    """components/pointnet2.py:44-52"""
    h = mlp(sd, prefix + ".nn", torch.cat([x, pos], dim=1))
    B = len(ptr) - 1
    out = torch.stack([h[ptr[b]:ptr[b + 1]].max(dim=0)[0] for b in range(B)])
    return out, pos.new_zeros((B, 3)), np.arange(B + 1, dtype=np.int64)


def fp_module(sd, prefix, x, pos, ptr, x_skip, pos_skip, ptr_skip, k):
    """components/pointnet2.py:70-76"""
    y = torch.from_numpy(O.knn_interpolate(x.numpy(), pos.numpy(), ptr, pos_skip.numpy(), ptr_skip, k))
    if x_skip is not None:
        y = torch.cat([y, x_skip], dim=1)
    return mlp(sd, prefix + ".nn", y), pos_skip, ptr_skip


def pointnet2_nocs_forward(sd, hp, x, pos, batch, prefix="pointnet2_nocs", return_intermediates=False):
    """networks/pointnet2_nocs.py:134-166 (eval mode: dropout = identity)."""
    p = hp
    ptr = O.batch_to_ptr(batch.numpy())
    sa0 = (x, pos, ptr)
    sa1 = sa_module(sd, prefix + ".sa1_module", *sa0, p["sa1_ratio"], p["sa1_r"], return_graph=True)
    g1 = sa1[3]
    sa1 = sa1[:3]
    sa2 = sa_module(sd, prefix + ".sa2_module", *sa1, p["sa2_ratio"], p["sa2_r"], return_graph=True)
    g2 = sa2[3]
    sa2 = sa2[:3]
    sa3 = global_sa_module(sd, prefix + ".sa3_module", *sa2)
    fp3 = fp_module(sd, prefix + ".fp3_module", *sa3, *sa2, p["fp3_k"])
    fp2 = fp_module(sd, prefix + ".fp2_module", *fp3, *sa1, p["fp2_k"])
    h, _, _ = fp_module(sd, prefix + ".fp1_module", *fp2, *sa0, p["fp1_k"])
    h = F.relu(F.linear(h, sd[prefix + ".lin1.weight"], sd[prefix + ".lin1.bias"]))
    features = F.linear(h, sd[prefix + ".lin2.weight"], sd[prefix + ".lin2.bias"])
    logits = F.linear(features, sd[prefix + ".lin3.weight"], sd[prefix + ".lin3.bias"])
    g = F.relu(sa3[0])
    g = F.linear(g, sd[prefix + ".global_lin1.weight"], sd[prefix + ".global_lin1.bias"])
    global_logits = F.linear(g, sd[prefix + ".global_lin2.weight"], sd[prefix + ".global_lin2.bias"])
    res = dict(per_point_features=features, per_point_logits=logits, per_point_batch_idx=batch,
               global_logits=global_logits, global_feature=sa3[0])
    if return_intermediates:
        res["_inter"] = dict(sa1_x=sa1[0], sa1_pos=sa1[1], sa1_idx=g1["idx"], sa1_nbr=g1["nbr"], sa1_cnt=g1["cnt"],
                             sa2_x=sa2[0], sa2_pos=sa2[1], sa2_idx=g2["idx"], sa2_nbr=g2["nbr"], sa2_cnt=g2["cnt"],
                             fp3_x=fp3[0], fp2_x=fp2[0], fp1_x=h)
    return res


def nocs_postprocess(logits, nocs_bins):
    """networks/conv_implicit_wnf.py:220-231: arg-max bin, soft-max confidence at it, bin -> coordinate."""
    lb = logits.reshape(logits.shape[0], nocs_bins, 3)
    idx = torch.argmax(lb, dim=1)
    conf = torch.squeeze(torch.gather(F.softmax(lb, dim=1), 1, idx.unsqueeze(1)))
    scales = (torch.tensor([1.0, 1.0, 1.0]) - torch.tensor([0.0, 0.0, 0.0])) / (torch.tensor([float(nocs_bins)] * 3) - 1)
    pred_nocs = idx * scales + torch.tensor([0.0, 0.0, 0.0])
    return idx, conf, pred_nocs


def pointnet2_forward(sd, hp, x, pos, batch):
    res = pointnet2_nocs_forward(sd, hp["pointnet2_params"], x, pos, batch)
    idx, conf, pred_nocs = nocs_postprocess(res["per_point_logits"], hp["pointnet2_params"]["nocs_bins"])
    res["nocs_data"] = dict(x=res["per_point_features"], pos=pred_nocs, batch=batch, sim_points=pos,
                            pred_confidence=conf, nocs_bin_idx=idx)
    return res


# ------------------------------------------------------------------------------------------------ gridding
def points_grid_idxs(points, lower, upper, grid_shape):
    """components/gridding.py:161-186 (float32 maths, truncation toward zero, clamp)."""
    lc = torch.tensor(lower, dtype=torch.float32)
    uc = torch.tensor(upper, dtype=torch.float32)
    idx_scale = torch.tensor(grid_shape, dtype=torch.float32) - 1
    f = (points + (-lc)) * (idx_scale / (uc - lc))
    i = f.to(torch.int64)
    for a in range(3):
        i[..., a] = torch.clamp(i[..., a], 0, grid_shape[a] - 1)
    return i


def idxs_to_points(idxs, lower, upper, grid_shape):
    """components/gridding.py:230-256"""
    lc = torch.tensor(lower, dtype=torch.float32)
    uc = torch.tensor(upper, dtype=torch.float32)
    idx_scale = torch.tensor(grid_shape, dtype=torch.float32) - 1
    return idxs * ((uc - lc) / idx_scale) + lc


def volume_agg(sd, hp, nocs, B, prefix="volume_agg", return_intermediates=False):
    """networks/conv_implicit_wnf.py:43-100 -> (B, C, G0, G1, G2)"""
    gs = tuple(hp["grid_shape"])
    pts = nocs["pos"]
    gi = points_grid_idxs(pts, hp["lower_corner"], hp["upper_corner"], gs)
    flat = ((nocs["batch"].to(torch.int64) * gs[0] + gi[:, 0]) * gs[1] + gi[:, 1]) * gs[2] + gi[:, 2]
    feats = [nocs["x"]]
    if hp.get("include_point_feature", True):
        feats.append(pts - idxs_to_points(gi, hp["lower_corner"], hp["upper_corner"], gs))
        feats.append(nocs["sim_points"])
    if hp.get("include_confidence_feature", False):
        feats.append(nocs["pred_confidence"])
    f = torch.cat(feats, dim=-1)
    f = mlp(sd, prefix + ".local_nn", f)
    C = f.shape[1]
    n = B * gs[0] * gs[1] * gs[2]
    red = {"max": "amax", "mean": "mean", "sum": "sum", "add": "sum"}[hp["reduce_method"]]
    vol = torch.zeros(C, n).scatter_reduce(1, flat.unsqueeze(0).expand(C, -1), f.t().contiguous(), red, include_self=False)
    vol = vol.reshape((C, B) + gs).permute(1, 0, 2, 3, 4).contiguous()
    if return_intermediates:
        return vol, dict(flat_idx=flat, point_features=f)
    return vol


# ------------------------------------------------------------------------------------------------ UNet
def unet_channel_plan(in_channels, f_maps, num_levels):
    """components/unet3d.py:127-144,416-433: [(name, cin, cout)] for all SingleConv layers."""
    if isinstance(f_maps, int):
        f_maps = [f_maps * 2 ** k for k in range(num_levels)]
    plan = []
    for i, fo in enumerate(f_maps):
        cin = in_channels if i == 0 else f_maps[i - 1]
        c1 = max(fo // 2, cin)
        plan.append((f"encoders.{i}", cin, c1, fo))
    rf = list(reversed(f_maps))
    for i in range(len(rf) - 1):
        plan.append((f"decoders.{i}", rf[i] + rf[i + 1], rf[i + 1], rf[i + 1]))
    return plan, f_maps


def _single_conv(sd, p, x, num_groups, order="gcr"):
    """components/unet3d.py:19-73 (create_conv) walked character by character: 'g' GroupNorm (over the channels it sees: the input's before the
    convolution, the output's behind it; one group when there are fewer channels than groups), 'b' BatchNorm3d (eval), 'c' Conv3d 3x3x3 pad 1
    (with a bias only when the order has no norm layer), 'r' ReLU, 'l' LeakyReLU(0.1), 'e' ELU."""
    assert "c" in order and order[0] not in "rle"
    for ch in order:
        if ch == "g":
            c = x.shape[1]
            g = num_groups if c >= num_groups else 1
            x = F.group_norm(x, g, sd[p + ".groupnorm.weight"], sd[p + ".groupnorm.bias"], eps=1e-5)
        elif ch == "b":
            x = F.batch_norm(x, sd[p + ".batchnorm.running_mean"], sd[p + ".batchnorm.running_var"], sd[p + ".batchnorm.weight"],
                             sd[p + ".batchnorm.bias"], training=False, eps=1e-5)
        elif ch == "c":
            x = F.conv3d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), padding=1)
        elif ch == "r":
            x = F.relu(x)
        elif ch == "l":
            x = F.leaky_relu(x, 0.1)
        elif ch == "e":
            x = F.elu(x)
        else:
            raise ValueError(f"unsupported layer type {ch!r}")
    return x


def unet3d(sd, hp, x, prefix="unet_3d.abstract_3d_unet", return_intermediates=False):
    """components/unet3d.py:449-474 with DoubleConv in any create_conv layer order (the shipped one: 'gcr')."""
    order = hp.get("layer_order", "gcr")
    ng = hp.get("num_groups", 8)
    nl = hp.get("num_levels", 4)
    f_maps = hp["f_maps"]
    n_enc = nl if isinstance(f_maps, int) else len(f_maps)
    skips = []
    inter = {}
    for i in range(n_enc):
        if i > 0:
            x = F.max_pool3d(x, 2)
        for j in (1, 2):
            x = _single_conv(sd, f"{prefix}.encoders.{i}.basic_module.SingleConv{j}", x, ng, order)
        inter[f"enc{i}"] = x
        skips.insert(0, x)
    skips = skips[1:]
    for i, s in enumerate(skips):
        x = F.interpolate(x, size=s.shape[2:], mode="nearest")
        x = torch.cat((s, x), dim=1)
        for j in (1, 2):
            x = _single_conv(sd, f"{prefix}.decoders.{i}.basic_module.SingleConv{j}", x, ng, order)
        inter[f"dec{i}"] = x
    x = F.conv3d(x, sd[prefix + ".final_conv.weight"], sd[prefix + ".final_conv.bias"])
    if return_intermediates:
        return x, inter
    return x


# ------------------------------------------------------------------------------------------------ decoder
def implicit_decoder(sd, prefix, vol, q):
    """networks/conv_implicit_wnf.py:128-149.  vol (B,C,D,H,W), q (B,M,3) -> (B,M,out)."""
    qn = 2.0 * q - 1.0
    s = F.grid_sample(vol, qn.view(*(qn.shape[:2] + (1, 1, 3))), mode="bilinear", padding_mode="border", align_corners=True)
    s = s.view(s.shape[:3]).permute(0, 2, 1)
    return mlp(sd, prefix + ".mlp", s)


def grid_points(Q):
    """components/gridding.py:139-159 with include_batch=False, unit cube, shape (Q,Q,Q,3) float32."""
    ar = torch.arange(Q, dtype=torch.int64)
    gi = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), dim=-1)
    scales = (torch.tensor([1.0] * 3) - torch.tensor([0.0] * 3)) / (torch.tensor([float(Q)] * 3) - 1)
    return gi.to(torch.float32) * scales + (-torch.tensor([0.0] * 3))


def decode_volume(sd, vol_b, Q, chunk=64, prefix="volume_decoder"):
    """predict.py:145-157 for one garment: vol_b (1,C,G,G,G) -> (Q,Q,Q) float32."""
    gp = grid_points(Q)
    out = torch.zeros(Q, Q, Q)
    for z0 in range(0, Q, chunk):
        for y0 in range(0, Q, chunk):
            for x0 in range(0, Q, chunk):
                sl = (slice(z0, min(Q, z0 + chunk)), slice(y0, min(Q, y0 + chunk)), slice(x0, min(Q, x0 + chunk)))
                qp = gp[sl]
                r = implicit_decoder(sd, prefix, vol_b, qp.reshape(1, -1, 3))
                out[sl] = r.view(*qp.shape[:-1])
    return out


# ------------------------------------------------------------------------------------------------ whole path
def isosurface(wnf, level, sigma, gradient_direction="ascent"):
    """predict.py:160-181 -> dict(verts f64 (V,3) in [0,1], faces, normals, values, verts_ggm)"""
    Q = wnf.shape[-1]
    g = O.ggm(wnf, sigma)
    spacing = 1 / (Q - 1)
    v, f, n, a = O.marching_cubes(wnf, level, (spacing,) * 3, gradient_direction)
    return dict(verts=v, faces=f, normals=n, values=a, verts_ggm=O.gather_nn(g, v, spacing), ggm=g)


def grip_postprocess(p2, pos, batch, b, nocs_bins):
    """predict.py:254-274 for garment b"""
    lg = p2["global_logits"][b:b + 1]
    bins = lg.reshape(1, nocs_bins, 3)
    idx = torch.argmax(bins, dim=1)
    scales = (torch.tensor([1.0] * 3) - torch.tensor([0.0] * 3)) / (torch.tensor([float(nocs_bins)] * 3) - 1)
    sel = batch == b
    dist = torch.norm(pos[sel], p=None, dim=1)
    return dict(pred_global_nocs_grip_point=(idx * scales + torch.tensor([0.0] * 3))[0], pred_global_confidence=torch.softmax(bins, dim=1)[0],
                pred_nocs_grip_point=p2["nocs_data"]["pos"][sel][torch.argmin(dist)], global_feature=p2["global_feature"][b])


def predict(sd, hp, x, pos, batch, Q=128, level=0.5, sigma=0.5, auto_level=False):
    """predict.py:138-209 for a batch; returns per-garment results (the reference asserts batch_size==1)."""
    with torch.no_grad():
        B = int(batch.max()) + 1
        p2 = pointnet2_forward(sd, hp, x, pos, batch)
        vol_in = volume_agg(sd, hp["volume_agg_params"], p2["nocs_data"], B)
        vol = unet3d(sd, hp["unet3d_params"], vol_in)
        outs = []
        for b in range(B):
            wnf = decode_volume(sd, vol[b:b + 1], Q).numpy()
            lv = 0.5 * (float(wnf.min()) + float(wnf.max())) if auto_level else level
            r = dict(wnf_volume=wnf, level=lv)
            r.update({k: v.numpy() for k, v in grip_postprocess(p2, pos, batch, b, hp["pointnet2_params"]["nocs_bins"]).items()})
            try:
                iso = isosurface(wnf, lv, sigma)
                sq = torch.from_numpy(iso["verts"].astype(np.float32)).view(1, -1, 3)
                iso["warp_field"] = implicit_decoder(sd, "surface_decoder", vol[b:b + 1], sq).view(-1, 3).numpy()
                r.update(iso)
            except ValueError:
                pass
            outs.append(r)
        return dict(pointnet2_result=p2, in_feature_volume=vol_in, out_feature_volume=vol, garments=outs)
