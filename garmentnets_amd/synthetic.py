"""Synthetic inputs for tests and benchmarks (no dataset / checkpoint is available offline).

* ``default_hparams``      -- the reference's shipped hyper-parameters
                              (config/train_pointnet2_default.yaml:30-48, config/train_pipeline_default.yaml:39-74).
* ``state_dict_spec``      -- (key, shape) list of the reference checkpoint schema, derived from
                              the hyper-parameters only.
* ``synthetic_state_dict`` -- seeded random weights; each tensor is drawn from its own generator keyed by the
                              parameter NAME, so the values do not depend on module construction order.
* ``synthetic_cloud``      -- seeded "dress" point clouds (open noisy cylinder) of SURVEY.md 8d.
"""
import copy
import zlib

import numpy as np
import torch


def default_hparams(grid=32, reduce_method="max", mc_surface=False):
    hp = {
        "pointnet2_params": dict(feature_dim=128, batch_norm=True, dropout=True, sa1_ratio=0.5, sa1_r=0.05,
                                 sa2_ratio=0.25, sa2_r=0.1, fp3_k=1, fp2_k=3, fp1_k=3, symmetry_axis=None, nocs_bins=64),
        "volume_agg_params": dict(nn_channels=[137, 137, 128], batch_norm=True, lower_corner=[0, 0, 0],
                                  upper_corner=[1, 1, 1], grid_shape=[grid, grid, grid], reduce_method=reduce_method,
                                  include_point_feature=True, include_confidence_feature=True),
        "unet3d_params": dict(in_channels=128, out_channels=128, f_maps=32, layer_order="gcr", num_groups=8, num_levels=4),
        "volume_decoder_params": dict(nn_channels=[128, 256, 256, 1], batch_norm=True),
        "surface_decoder_params": dict(nn_channels=[128, 256, 256, 3], batch_norm=True),
        "mc_surface_decoder_params": dict(nn_channels=[128, 256, 256, 1], batch_norm=True),
        "mc_surface_loss_weight": 1 if mc_surface else 0,
    }
    return copy.deepcopy(hp)


def _mlp_spec(prefix, channels, batch_norm=True):
    out = []
    for i in range(1, len(channels)):
        p = f"{prefix}.{i - 1}"
        out.append((p + ".0.weight", (channels[i], channels[i - 1])))
        out.append((p + ".0.bias", (channels[i],)))
        if batch_norm:
            out += [(p + ".2.weight", (channels[i],)), (p + ".2.bias", (channels[i],)),
                    (p + ".2.running_mean", (channels[i],)), (p + ".2.running_var", (channels[i],)),
                    (p + ".2.num_batches_tracked", ())]
    return out


def unet_plan(in_channels, f_maps, num_levels):
    """[(module prefix, conv1 (cin,cout), conv2 (cin,cout))] following components/unet3d.py:127-144,416-433."""
    if isinstance(f_maps, int):
        f_maps = [f_maps * 2 ** k for k in range(num_levels)]
    plan = []
    for i, fo in enumerate(f_maps):
        cin = in_channels if i == 0 else f_maps[i - 1]
        c1 = max(fo // 2, cin)
        plan.append((f"encoders.{i}", (cin, c1), (c1, fo)))
    rf = list(reversed(f_maps))
    for i in range(len(rf) - 1):
        plan.append((f"decoders.{i}", (rf[i] + rf[i + 1], rf[i + 1]), (rf[i + 1], rf[i + 1])))
    return plan, f_maps


def pointnet2_spec(p, prefix="pointnet2_nocs"):
    bn = p.get("batch_norm", True)
    fd = p["feature_dim"]
    od = 3 if p.get("nocs_bins") is None else p["nocs_bins"] * 3
    s = []
    s += _mlp_spec(prefix + ".sa1_module.conv.local_nn", [6, 64, 64, 128], bn)
    s += _mlp_spec(prefix + ".sa2_module.conv.local_nn", [131, 128, 128, 256], bn)
    s += _mlp_spec(prefix + ".sa3_module.nn", [259, 256, 512, 1024], bn)
    s += _mlp_spec(prefix + ".fp3_module.nn", [1280, 256, 256], bn)
    s += _mlp_spec(prefix + ".fp2_module.nn", [384, 256, 128], bn)
    s += _mlp_spec(prefix + ".fp1_module.nn", [131, 128, 128, 128], bn)
    for name, (o, i) in (("lin1", (128, 128)), ("lin2", (fd, 128)), ("lin3", (od, fd)),
                         ("global_lin1", (1024, 1024)), ("global_lin2", (od, 1024))):
        s += [(f"{prefix}.{name}.weight", (o, i)), (f"{prefix}.{name}.bias", (o,))]
    return s


def state_dict_spec(hp):
    s = pointnet2_spec(hp["pointnet2_params"])
    va = hp["volume_agg_params"]
    s += _mlp_spec("volume_agg.local_nn", va["nn_channels"], va.get("batch_norm", True))
    u = hp["unet3d_params"]
    plan, f_maps = unet_plan(u["in_channels"], u["f_maps"], u.get("num_levels", 4))
    up = "unet_3d.abstract_3d_unet"
    for mod, c1, c2 in plan:
        for j, (ci, co) in ((1, c1), (2, c2)):
            p = f"{up}.{mod}.basic_module.SingleConv{j}"
            s += [(p + ".groupnorm.weight", (ci,)), (p + ".groupnorm.bias", (ci,)), (p + ".conv.weight", (co, ci, 3, 3, 3))]
    s += [(up + ".final_conv.weight", (u["out_channels"], f_maps[0], 1, 1, 1)), (up + ".final_conv.bias", (u["out_channels"],))]
    decs = [("volume_decoder", hp["volume_decoder_params"]), ("surface_decoder", hp["surface_decoder_params"])]
    if hp.get("mc_surface_loss_weight", 0) > 0:
        decs.append(("mc_surface_decoder", hp["mc_surface_decoder_params"]))
    for name, dp in decs:
        s += _mlp_spec(name + ".mlp", list(dp["nn_channels"]), dp.get("batch_norm", True))
    return s


def synthetic_tensor(key, shape, seed=0):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    if key.endswith("num_batches_tracked"):
        return torch.tensor(1000, dtype=torch.int64)
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("running_var"):
        return torch.rand(shape, generator=g) + 0.5
    if key.endswith(".2.weight") or key.endswith("groupnorm.weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if key.endswith(".2.bias") or key.endswith("groupnorm.bias"):
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith("weight"):
        fan_in = int(np.prod(shape[1:]))
        bound = (3.0 / fan_in) ** 0.5  # unit-gain uniform: keeps activations O(1) through the stack
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if key.endswith("bias"):
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.1
    raise KeyError(key)


def synthetic_state_dict(hp, seed=0, planted_nocs=False, planted_wnf=False):
    """planted_nocs: see plant_nocs_path; planted_wnf: see plant_wnf_path (True = its defaults, or a dict of its keyword arguments) --
    bench.py's default weights carry both"""
    sd = {k: synthetic_tensor(k, shp, seed) for k, shp in state_dict_spec(hp)}
    if planted_nocs:
        sd = plant_nocs_path(sd, hp)
    if planted_wnf:
        sd = plant_wnf_path(sd, hp, **(planted_wnf if isinstance(planted_wnf, dict) else {}))
    return sd


NOCS_PLANT_GAP = 8.0      # logit margin between the planted arg-max bin and its neighbours


def plant_nocs_path(sd, hp, prefix="pointnet2_nocs"):
    """Seeded random weights predict the SAME few NOCS bins for every point of a cloud (the 6000 points collapse into ~5 of the 2 M cells of a
    128^3 grid): a degenerate input for everything behind the gridding (GroupNorm over a > 99.99 % empty volume amplifies rounding ~100x)
    and nothing like what a trained PointNet++ produces.  This plants ONE trained-like behaviour into the random checkpoint, inside the
    reference's own schema (the result loads into the reference's modules unchanged): the NOCS head reads the per-point input feature
    ``x`` (the cloud's colour channels, networks/pointnet2_nocs.py:145-157 -- the only absolute per-point quantity the architecture
    carries to the head) through an identity path

        fp1_module.nn.{0,1,2} channel j <- x_j   (Linear row = unit vector, bias 0, BatchNorm = identity)
        lin1, lin2            channel j <- channel j
        lin3                  logit (bin k, axis a) += s (2 c_a k/63 - (k/63)^2)    [= -s (c_a - k/63)^2 + const: arg-max = nearest bin]

    on top of the random weights of all other channels (which keep reading the planted ones).  With ``synthetic_cloud(colour="position")``
    -- x := the garment's normalised, 64-bin quantised point positions -- the predicted NOCS coordinates spread over the garment's own
    shape (thousands of occupied cells); the margin to the neighbouring bins is NOCS_PLANT_GAP logits, the random channels add O(1) on
    top, so the soft-max confidence stays a non-trivial per-point quantity (0.99-0.9999)."""
    bins = hp["pointnet2_params"]["nocs_bins"]
    assert bins is not None and hp["pointnet2_params"].get("batch_norm", True)
    sd = dict(sd)
    ident = torch.eye(3)
    for i, col0 in ((0, 128), (1, 0), (2, 0)):
        p = f"{prefix}.fp1_module.nn.{i}"
        w = sd[p + ".0.weight"].clone()
        w[:3] = 0.0
        w[:3, col0:col0 + 3] = ident
        sd[p + ".0.weight"] = w
        for key, val in ((".0.bias", 0.0), (".2.weight", 1.0), (".2.bias", 0.0), (".2.running_mean", 0.0), (".2.running_var", 1.0)):
            t = sd[p + key].clone()
            t[:3] = val
            sd[p + key] = t
    for name in ("lin1", "lin2"):
        w, b = sd[f"{prefix}.{name}.weight"].clone(), sd[f"{prefix}.{name}.bias"].clone()
        w[:3] = 0.0
        w[:3, :3] = ident
        b[:3] = 0.0
        sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"] = w, b
    s = NOCS_PLANT_GAP * float(bins - 1) ** 2
    w, b = sd[f"{prefix}.lin3.weight"].clone(), sd[f"{prefix}.lin3.bias"].clone()
    w[:, :3] = 0.0
    k = torch.arange(bins, dtype=torch.float64) / (bins - 1)
    for a in range(3):                                  # logits are laid out (bin, axis): networks/conv_implicit_wnf.py:220-224
        w[a::3, a] = (2.0 * s * k).float()
        b[a::3] = b[a::3] - (s * k * k).float()
    sd[f"{prefix}.lin3.weight"], sd[f"{prefix}.lin3.bias"] = w, b
    return sd


WNF_PLANT = dict(gain=1.0, offset=0.1, cut=0.35, ramp=8.0, noise=0.01)      # tuned on the 128^3 / mean benchmark configuration


def plant_wnf_path(sd, hp, gain=None, offset=None, cut=None, ramp=None, noise=None):
    """Seeded random weights make the winding-number field random-weight noise: its 0.5 level set is a sponge of ~500 k vertices per
    garment, ten times a real garment's 40-60 k, so everything behind the lattice decoder (marching cubes, the surface decoder, the D2H
    copy of the mesh) is timed on the wrong amount of work.  This plants, inside the reference's schema and next to the random weights, a
    CARRIER path whose output is a smooth field that is high near the garment and zero away from it -- the level set becomes a thin shell
    hugging (part of) the garment's surface:

      volume_agg.local_nn (last layer)   channels of GroupNorm group 0 of the UNet's input: ReLU(ramp (x_0 - cut)) clamped to 1 by the
                                         following layer's normalisation -- a per-point amplitude that switches the carrier on over the part
                                         of the garment with planted colour / position coordinate x_0 > cut (point feature 0 = the planted
                                         NOCS path's x_0; without plant_nocs_path it is a random feature and the shell is simply irregular)
      every UNet convolution             carrier rows: a 3x3x3 box average over the carrier channels of its input (GroupNorm group 0; the
                                         decoders' first convolutions also read the LAST group, which lies in the upsampled source), zero
                                         on every other input channel; GroupNorm affine of carrier channels = identity.  Level by level this
                                         is a multi-scale blur of the occupancy, renormalised by each GroupNorm; the ReLUs keep it >= 0 and
                                         exactly 0 where the volume is at rest
      final 1x1x1 convolution            output channel 0 = mean of the last layer's carrier channels
      volume_decoder.mlp                 hidden unit 0 of both hidden layers passes feature 0 through; the output layer is
                                         gain * carrier + offset + noise * (the random hidden units' contribution, unit variance)
                                         and its BatchNorm is the identity, so WNF = ReLU(that)

    The other output rows of every layer keep their random weights (and read the carrier channels like any other input): the kernels do
    the same work on the same shapes.  The surface (warp-field) decoder is left random."""
    cfg = dict(WNF_PLANT)
    cfg.update({k: v for k, v in dict(gain=gain, offset=offset, cut=cut, ramp=ramp, noise=noise).items() if v is not None})
    sd = dict(sd)
    u = hp["unet3d_params"]
    groups = u.get("num_groups", 8)
    plan, f_maps = unet_plan(u["in_channels"], u["f_maps"], u.get("num_levels", 4))
    up = "unet_3d.abstract_3d_unet"
    nenc = len(f_maps)

    def group(cin, g):
        gs = cin // groups
        return list(range(g * gs, (g + 1) * gs))

    def plant_conv(prefix, cin, carriers_in, carriers_out):
        w = sd[prefix + ".conv.weight"].clone()
        w[carriers_out] = 0.0
        row = torch.zeros(cin, 3, 3, 3)
        row[carriers_in] = 1.0 / (27.0 * len(carriers_in))
        w[carriers_out] = row
        sd[prefix + ".conv.weight"] = w
        gw, gb = sd[prefix + ".groupnorm.weight"].clone(), sd[prefix + ".groupnorm.bias"].clone()
        gw[carriers_in], gb[carriers_in] = 1.0, 0.0
        sd[prefix + ".groupnorm.weight"], sd[prefix + ".groupnorm.bias"] = gw, gb

    # what the consumers of every module's OUTPUT read as carriers
    def out_carriers(k):
        mod, c1, c2 = plan[k]
        cout = c2[1]
        if k < nenc - 1:                                  # encoder: next encoder's first conv + its decoder's skip group 0
            nxt = group(plan[k + 1][1][0], 0)
            skip_cin = plan[2 * nenc - 2 - k][1][0]       # decoder that takes this encoder's output as skip
            return sorted(set(nxt) | set(group(skip_cin, 0)))
        if k < len(plan) - 1:                             # deepest encoder / a decoder: the next decoder's LAST group, in the upsampled part
            cat_cin = plan[k + 1][1][0]
            skip_c = cat_cin - cout
            return [c - skip_c for c in group(cat_cin, groups - 1)]
        return group(cout, 0)                             # last decoder -> final conv

    for k, (mod, c1, c2) in enumerate(plan):
        cin = c1[0]
        cin_carriers = group(cin, 0)
        if k >= nenc:                                     # decoder: concatenated (skip, upsampled) input
            cin_carriers = cin_carriers + group(cin, groups - 1)
        mid = group(c1[1], 0)
        plant_conv(f"{up}.{mod}.basic_module.SingleConv1", cin, cin_carriers, mid)
        plant_conv(f"{up}.{mod}.basic_module.SingleConv2", c2[0], mid, out_carriers(k))
    # final 1x1x1: output channel 0 <- mean of the last decoder's carriers
    fw, fb = sd[up + ".final_conv.weight"].clone(), sd[up + ".final_conv.bias"].clone()
    last = group(f_maps[0], 0)
    fw[0] = 0.0
    fw[0, last] = 1.0 / len(last)
    fb[0] = 0.0
    sd[up + ".final_conv.weight"], sd[up + ".final_conv.bias"] = fw, fb
    # the UNet's input carriers: volume_agg's last layer, amplitude ReLU(ramp (x_0 - cut)) of point feature 0
    va = hp["volume_agg_params"]
    nl = len(va["nn_channels"]) - 2
    p = f"volume_agg.local_nn.{nl}"
    in_carriers = group(u["in_channels"], 0)
    if nl == 0:
        w, b = sd[p + ".0.weight"].clone(), sd[p + ".0.bias"].clone()
        w[in_carriers] = 0.0
        w[in_carriers, 0] = cfg["ramp"]
        b[in_carriers] = -cfg["ramp"] * cfg["cut"]
    else:                                                 # hidden layer first: pass point feature 0 through hidden unit 0 (>= 0: a planted colour in [0.1, 0.9])
        p0 = f"volume_agg.local_nn.{nl - 1}"
        w0, b0 = sd[p0 + ".0.weight"].clone(), sd[p0 + ".0.bias"].clone()
        w0[0] = 0.0
        w0[0, 0] = 1.0
        b0[0] = 0.0
        sd[p0 + ".0.weight"], sd[p0 + ".0.bias"] = w0, b0
        _identity_bn(sd, p0, [0], va.get("batch_norm", True))
        w, b = sd[p + ".0.weight"].clone(), sd[p + ".0.bias"].clone()
        w[in_carriers] = 0.0
        w[in_carriers, 0] = cfg["ramp"]
        b[in_carriers] = -cfg["ramp"] * cfg["cut"]
    sd[p + ".0.weight"], sd[p + ".0.bias"] = w, b
    _identity_bn(sd, p, in_carriers, va.get("batch_norm", True))
    # the WNF decoder: feature 0 through hidden unit 0 of both hidden layers, then gain * carrier + offset (+ a little of everything else)
    dch = list(hp["volume_decoder_params"]["nn_channels"])
    dbn = hp["volume_decoder_params"].get("batch_norm", True)
    for i in range(len(dch) - 2):
        q = f"volume_decoder.mlp.{i}"
        w, b = sd[q + ".0.weight"].clone(), sd[q + ".0.bias"].clone()
        w[0] = 0.0
        w[0, 0] = 1.0
        b[0] = 0.0
        sd[q + ".0.weight"], sd[q + ".0.bias"] = w, b
        _identity_bn(sd, q, [0], dbn)
    q = f"volume_decoder.mlp.{len(dch) - 2}"
    w, b = sd[q + ".0.weight"].clone(), sd[q + ".0.bias"].clone()
    rest = w[:, 1:]
    w[:, 1:] = rest / rest.norm(dim=1, keepdim=True).clamp_min(1e-12) * cfg["noise"]     # unit-variance hidden units -> std `noise`
    w[:, 0] = cfg["gain"]
    b[:] = cfg["offset"]
    sd[q + ".0.weight"], sd[q + ".0.bias"] = w, b
    _identity_bn(sd, q, list(range(dch[-1])), dbn)
    return sd


def _identity_bn(sd, layer_prefix, channels, batch_norm=True):
    if not batch_norm:
        return
    for key, val in ((".2.weight", 1.0), (".2.bias", 0.0), (".2.running_mean", 0.0), (".2.running_var", 1.0)):
        t = sd[layer_prefix + key].clone()
        t[channels] = val
        sd[layer_prefix + key] = t


def synthetic_cloud(num_garments, n_points=6000, seed=0, first=0, colour="uniform"):
    """-> x (N,3) rgb in [0,1], pos (N,3) metres in the gripper frame, batch (N,) int64 sorted.
    colour: "uniform" = rgb ~ U(0,1); "position" = rgb := the garment's own normalised point positions, quantised to the
    64 NOCS bins in [0.1, 0.9] (a garment whose texture encodes where on the garment a point lies -- the input plant_nocs_path's NOCS
    head decodes, bench.py's default).
    Garment g of the seed's stream depends on (seed, g) only: ``first`` selects garments first..first+num_garments-1, i.e. exactly the
    slice [first, first+num) of the global batch synthetic_cloud(total, n, seed) (a rank's shard of it, batch ids restarting at 0)."""
    xs, ps, bs = [], [], []
    for b in range(num_garments):
        rng = np.random.Generator(np.random.PCG64(seed * 1000003 + first + b))
        while True:
            th = rng.uniform(0, 2 * np.pi, n_points)
            h = rng.uniform(0, 0.8, n_points)
            r = 0.25 + 0.03 * np.sin(3 * th) + rng.normal(0, 0.005, n_points)
            pos = np.stack([r * np.cos(th), r * np.sin(th), -h], axis=1).astype(np.float32)
            if len(np.unique(pos, axis=0)) == n_points:
                break
        rgb = rng.uniform(0, 1, (n_points, 3)).astype(np.float32)
        if colour == "position":
            mn, mx = pos.min(axis=0, keepdims=True), pos.max(axis=0, keepdims=True)
            rgb = (np.round((0.1 + 0.8 * (pos - mn) / (mx - mn)) * 63.0) / 63.0).astype(np.float32)
        elif colour != "uniform":
            raise ValueError(f"colour={colour!r}")
        xs.append(rgb)
        ps.append(pos)
        bs.append(np.full(n_points, b, np.int64))
    return (torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ps)), torch.from_numpy(np.concatenate(bs)))


def shell_volume(Q):
    """Analytic WNF-like shell used to benchmark / test the isosurface stage."""
    ax = (np.arange(Q, dtype=np.float64) / (Q - 1))
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    rho = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2)
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    return (sig(80 * (0.3 - rho)) * sig(80 * (0.4 - np.abs(Z - 0.5)))).astype(np.float32)
