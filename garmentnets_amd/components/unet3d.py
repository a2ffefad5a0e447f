"""3-D UNet (GroupNorm -> Conv3d -> ReLU double convs, max-pool encoder, nearest-upsample + concat decoder) on HIP.

Checkpoint-schema twin of the subset of /root/reference/components/unet3d.py that the pipeline instantiates
(``Abstract3DUNet`` with ``basic_module=DoubleConv``, ``layer_order='gcr'``): module / parameter names
``encoders.{i}.basic_module.SingleConv{1,2}.{groupnorm,conv}``, ``decoders.{i}...``, ``final_conv`` are kept.
The torch layers only hold parameters.  Execution is channel-last ([B][D][H][W][C]):
    GroupNorm  = gn_channel_stats (+ gn_groupnorm_affine)  -> per-(sample, channel) affine
    conv+ReLU  = gn_conv3d_gcr: the affine is applied while the input halo is staged into LDS, the 3x3x3 conv runs as
                 an implicit GEMM on fp32 MFMA; nearest upsampling and torch.cat((skip, x)) of the decoder
                 (components/unet3d.py:291,330) are folded into the loader (second source read at half resolution)
    MaxPool3d  = gn_maxpool3d_2,   final 1x1x1 conv = gn_linear.
"""
import torch
from torch import nn

from .. import arith as AR
from .. import ops
from .mlp import PackedModule, pack_wb, param_cache


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


def _wino_ok(arith, src0, cout):
    """the layer fits a Winograd F(2,3)-along-x kernel (arith.winograd: csrc/unet_wino.hip for Cout % 128 == 0; arith.winograd32: csrc/unet_wino32.hip for the
    32- / 64-wide layers).  Decided from the SAMPLE's shape alone, never
    from the batch size: the Winograd and the direct form round differently, and a garment's result must not depend on how many garments share its
    batch (the direct kernels may switch variant with the batch size because they are bit-identical to each other)"""
    _, D, H, W, cin = src0.shape
    if not (arith.winograd and arith.conv_mode == ops.SPLIT_F16X2 and ops.wino_supported(cin, cout, (D, H, W))):
        return False
    if cout % 128 == 0:
        return (D // 4) * (H // 8) * (W // 8) * (cout // 128) >= 32
    # the 32-wide column-block kernel (csrc/unet_wino32.hip, round 6): 8 x 8 x 8 tiles; one workgroup per CU walks chains of tiles -- worth it from a
    # few tiles per CU and sample on (the 64^3 and 128^3 levels of the UNet)
    return arith.winograd32 and (D // 8) * (H // 8) * (W // 8) * (cout // 32) >= 512


_ACT = {"r": ("ReLU", lambda: nn.ReLU(inplace=True), ops.ACT_RELU), "l": ("LeakyReLU", lambda: nn.LeakyReLU(negative_slope=0.1, inplace=True), ops.ACT_LEAKY),
        "e": ("ELU", lambda: nn.ELU(inplace=True), ops.ACT_ELU)}


def _class_constants(small_out, reach, cout):
    """border-class constants of an occupancy-aware launch from the layer's output over a small all-at-rest volume (n^3, n = 5 for the direct
    kernels, 8 -- the smallest volume of whole tiles -- for the Winograd kernel): per axis the voxels at distance 0 [, 1] from either face and one
    in the middle.  (Winograd along x: voxel 0 / n-1 of the small volume has the parity of voxel 0 / W-1 of the real one, n and W being multiples
    of 8, and away from the cells both members of an output pair hold the same value -- the differences d0 - d2, d2 - d1, d1 - d3 are exactly zero.)
    Strided / indexed views of a device tensor: nothing comes from the host, the path can be captured into a HIP graph."""
    B, n = small_out.shape[0], small_out.shape[1]
    pos = [0, n // 2, n - 1] if reach == 1 else [0, 1, n // 2, n - 2, n - 1]
    k = small_out
    for dim in (1, 2, 3):                          # (narrow + cat: device-side copies only -- an index tensor would be a host-to-device copy)
        k = torch.cat([k.narrow(dim, q, 1) for q in pos], dim=dim)
    return k.reshape(B, len(pos) ** 3, cout).contiguous()


class SingleConv(PackedModule, nn.Sequential):
    """One conv block of the reference's create_conv (components/unet3d.py:19-73), same module names and parameter layout for every layer order:
    'g' GroupNorm, 'b' BatchNorm3d, 'c' Conv3d 3x3x3 pad 1 (bias only when the order has no norm layer), 'r' ReLU, 'l' LeakyReLU(0.1), 'e' ELU.
    'gcr' -- what the GarmentNets pipeline ships -- runs on the fused split-operand kernels (run()); every other order ('cr', 'crg', 'cl', 'ce',
    'bcr', ...) runs the convolution on the exact fp32-MFMA kernel with the pre-conv normalisation applied on load and the rest (bias, LeakyReLU / ELU,
    a normalisation behind the non-linearity) through gn_affine_act: correct for any checkpoint, not tuned."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="gcr", num_groups=8, padding=1):
        super().__init__()
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError("garmentnets_amd implements the 3x3x3, padding 1 SingleConv of the GarmentNets pipeline")
        if "c" not in order or order.count("c") != 1:
            raise ValueError("Conv layer MUST be present (exactly once)")
        if order[0] in "rle":
            raise ValueError("Non-linearity cannot be the first operation in the layer")
        if any(ch not in "bgrlec" for ch in order):
            raise ValueError(f"Unsupported layer type in {order!r}. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c']")
        if any(ch in "rle" for ch in order[:order.index("c")]) or sum(ch in "gb" for ch in order[:order.index("c")]) > 1:
            raise NotImplementedError(f"layer order {order!r}: at most one normalisation and no non-linearity in front of the convolution")
        self.order = order
        for i, ch in enumerate(order):
            before = i < order.index("c")
            nch = in_channels if before else out_channels
            if ch in _ACT:
                self.add_module(_ACT[ch][0], _ACT[ch][1]())
            elif ch == "c":
                self.add_module("conv", nn.Conv3d(in_channels, out_channels, 3, padding=1, bias=not ("g" in order or "b" in order)))
            elif ch == "g":
                groups = num_groups if nch >= num_groups else 1
                assert nch % groups == 0, f"Expected number of channels in input to be divisible by num_groups. num_channels={nch}, num_groups={groups}"
                self.add_module("groupnorm", nn.GroupNorm(num_groups=groups, num_channels=nch))
            elif ch == "b":
                self.add_module("batchnorm", nn.BatchNorm3d(nch))

    def _pack(self):
        wp = ops.pack_conv_weight(self.conv.weight)                # [tap][Cin/16][Cout][16]
        gn = getattr(self, "groupnorm", None)
        return wp, (None if gn is None else gn.weight.detach().float().contiguous()), (None if gn is None else gn.bias.detach().float().contiguous())

    def _norm_affine(self, ch, B, st0, st1):
        """per-(sample, channel) affine of one normalisation layer from the statistics of what it normalises"""
        if ch == "g":
            return ops.groupnorm_affine(st0, st1, self.groupnorm.num_groups, self.groupnorm.eps, self.groupnorm.weight.detach().float().contiguous(),
                                        self.groupnorm.bias.detach().float().contiguous())
        bn = self.batchnorm                       # eval BatchNorm3d: running statistics
        sc = (bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps))
        sh = bn.bias.detach().double() - bn.running_mean.double() * sc
        return sc.float().expand(B, -1).contiguous(), sh.float().expand(B, -1).contiguous()

    def _run_generic(self, src0, src1, stats0, stats1, with_stats):
        """every layer order but 'gcr' (see the class docstring)"""
        order, ic = self.order, self.order.index("c")
        B, cout = src0.shape[0], self.conv.out_channels
        cin = src0.shape[-1] + (0 if src1 is None else src1.shape[-1])
        wp, _, _ = self.packed()
        if ic == 0:
            a = torch.ones((B, cin), dtype=torch.float32, device=src0.device)
            d = torch.zeros_like(a)
        else:
            st0 = st1 = None
            if order[0] == "g":
                st0 = stats0 if stats0 is not None else ops.channel_stats(src0)
                st1 = None if src1 is None else (stats1 if stats1 is not None else ops.channel_stats(src1))
            a, d = self._norm_affine(order[0], B, st0, st1)
        post = order[ic + 1:]
        bias = None if self.conv.bias is None else self.conv.bias.detach().float().contiguous()
        fuse_relu = bias is None and post[:1] == "r"
        y = ops.conv3d_gcr(src0, src1, a, d, wp, cout, relu=fuse_relu)
        k = 1 if fuse_relu else 0
        if bias is not None:                      # conv bias, fused with the non-linearity that follows it (if one does)
            act = _ACT[post[0]][2] if post[:1] and post[0] in _ACT else ops.ACT_NONE
            ops.affine_act(y, bias=bias, act=act, out=y)
            k = 1 if act != ops.ACT_NONE else 0
        for ch in post[k:]:
            if ch in _ACT:
                ops.affine_act(y, act=_ACT[ch][2], out=y)
            else:
                na, nd = self._norm_affine(ch, B, ops.channel_stats(y) if ch == "g" else None, None)
                ops.affine_act(y, a=na, d=nd, out=y)
        return y, (ops.channel_stats(y) if with_stats else None)

    def run(self, src0, src1=None, stats0=None, stats1=None, with_stats=True, sparse=None, arith=None, rest0=None):
        """arith: the arith.Arith of this call (None: arith.DEFAULT).  src0 [B][D][H][W][C0] (full res), src1 [B][D/2][H/2][W/2][C1] or None -> ([B][D][H][W][Cout], output stats).
        stats0/stats1: (sum, sumsq, V) of the inputs when the producing kernel already emitted them.
        sparse: occupancy-aware launch (split-operand modes), a dict
            flat      flat cell index of every point gn_grid_scatter scattered into the volume this layer descends from
            reach     1: src0 IS that volume; 2: src0 is the output of the reach-1 layer (non-constant within one voxel of the cells)
            small_in  (reach 2) the reach-1 layer's output over the 5 x 5 x 5 all-zero volume; this call adds 'small_out', its own
            rest_in   (reach 2) [B][C0]: the value src0 holds away from the cells; this call adds 'rest_out', its own output's
        arith.sparse_first_conv: only the output tiles that can see an occupied cell (within `reach`) go through the matrix cores; the rest
        are border-class constants taken from a dense launch of this layer over the 5^3 zero volume with the same affine: bit-identical output.
        arith.affine_in_weights (f16x2): the GroupNorm affine moves into per-sample weights and a bias table (ops.conv_affine_pack), so that
        the matrix cores multiply exact zeros wherever src0 is at rest -- same MACs, less power, more clock (csrc/conv_prep.hip).
        rest0 [B][C0]: (polyphase form of a decoder layer) the value the skip connection src0 holds away from the cells: its full-resolution
        launch takes the affine-in-weights form as well."""
        arith = arith or AR.DEFAULT
        if self.order != "gcr":
            return self._run_generic(src0, src1, stats0, stats1, with_stats)
        wp, gamma, beta = self.packed()
        st0 = stats0 if stats0 is not None else ops.channel_stats(src0)
        st1 = None
        if src1 is not None:
            st1 = stats1 if stats1 is not None else ops.channel_stats(src1)
        if (sparse is not None and arith.affine_in_weights and arith.conv_mode == ops.SPLIT_F16X2 and src1 is None
                and src0.shape[-1] % 16 == 0 and self.conv.out_channels % 32 == 0
                and (sparse["reach"] == 1 or sparse.get("rest_in") is not None)):
            return self._run_at_rest(src0, st0, with_stats, sparse, arith, gamma, beta)
        if arith.conv_mode != ops.CONV_FP32:      # split-operand path on the 16-bit matrix cores (csrc/unet_split.hip)
            mode = arith.conv_mode
            # fp16 planes: the sample's activations are range-normalised by a power of two (exact, undone in the epilogue)
            a, d, act_inv = ops.groupnorm_affine(st0, st1, self.groupnorm.num_groups, self.groupnorm.eps, gamma, beta, with_act_scale=True) \
                if mode == ops.SPLIT_F16X2 else ops.groupnorm_affine(st0, st1, self.groupnorm.num_groups, self.groupnorm.eps, gamma, beta) + (None,)
            cache, gen = param_cache(self, "_split_packs"), (self.conv.weight.device, self.conv.weight._version)
            wpack = cache.get(gen, mode, lambda: ops.pack_conv_weight_split(self.conv.weight, mode).to(self.conv.weight.device))
            cout, sp = self.conv.out_channels, {}
            wino = src1 is None and _wino_ok(arith, src0, cout)
            wwino = cache.get(gen, "wino", lambda: ops.pack_conv_weight_split_wino(self.conv.weight).to(self.conv.weight.device)) if wino else None
            if (sparse is not None and arith.sparse_first_conv and src1 is None and mode != ops.SPLIT_BF16X3 and src0.shape[-1] <= 384
                    and (cout % 128 == 0 or cout % 64 != 0) and min(src0.shape[1:4]) > 2 * sparse["reach"]
                    and (sparse["reach"] == 1 or sparse.get("small_in") is not None)
                    # (the class constants must come from the kernel the real launch takes: a Winograd layer handed a small volume it cannot take
                    #  -- 5^3, from a layer that ran in the direct form -- simply runs dense)
                    and (not wino or sparse.get("small_in") is None or ops.wino_supported(src0.shape[-1], cout, sparse["small_in"].shape[1:4]))):
                B, reach = src0.shape[0], int(sparse["reach"])
                small_in = sparse.get("small_in")
                if small_in is None:
                    n = 8 if wino else 5          # the Winograd kernel takes whole 4 x 8 x 8 tiles
                    small_in = torch.zeros((B, n, n, n, src0.shape[-1]), dtype=torch.float32, device=src0.device)
                # (a plain dense launch of the kernel the real launch takes)
                if wino:
                    small_out = ops.conv3d_gcr_split_wino(small_in, a, d, wwino, cout, relu=True, act_inv=act_inv)
                else:
                    small_out = ops.conv3d_gcr_split(small_in, None, a, d, wpack, cout, relu=True, act_inv=act_inv)
                sp = dict(tile_active=ops.grid_tile_flags(sparse["flat"], B, src0.shape[1:4], reach), kconst=_class_constants(small_out, reach, cout), kreach=reach)
                sparse["small_out"] = small_out
            if src1 is not None and arith.polyphase_upconv and mode != ops.SPLIT_BF16X3 and src1.shape[-1] <= 384:
                # polyphase form: the nearest-upsampled channels as a 2x2x2-tap convolution per output parity class on the COARSE volume
                # (8/27 of their MACs, the coarse halo staged once for all classes: csrc/upconv.hip), added in the fine launch's epilogue
                c0 = src0.shape[-1]
                def build_poly():
                    w0, wm, _ = ops.polyphase_weights(self.conv.weight, c0)
                    dev = self.conv.weight.device
                    return (ops.pack_conv_weight_split(w0, mode).to(dev), ops.pack_upconv_weight(wm, cout, mode).to(dev))
                pk0, pkm = cache.get(gen, ("poly", mode, c0), build_poly)
                part = ops.upconv_partial(src1, a[:, c0:].contiguous(), d[:, c0:].contiguous(), pkm, cout, act_inv=act_inv)
                # the full-resolution part in Winograd form (32- / 64-wide layers: csrc/unet_wino32.hip takes the partial in its epilogue)
                wino0 = mode == ops.SPLIT_F16X2 and cout % 128 != 0 and _wino_ok(arith, src0, cout)
                if rest0 is not None and arith.affine_in_weights and mode == ops.SPLIT_F16X2 and c0 % 16 == 0 and cout % 32 == 0:
                    # (a, d carry the sample's power-of-two activation scale: exact to undo)
                    a0 = (a[:, :c0] * act_inv[:, None]).contiguous()
                    d0 = (d[:, :c0] * act_inv[:, None]).contiguous()
                    w0c = cache.get(gen, ("w0", c0), lambda: self.conv.weight.detach()[:, :c0].contiguous())
                    prep = ops.conv_affine_pack(w0c, a0, d0, st0, rest0, wino=wino0)
                    r = ops.conv3d_gcr_split_persample(src0, prep, relu=True, with_stats=with_stats, partial=part)
                    return r if with_stats else (r, None)
                if wino0:
                    pkw = cache.get(gen, ("poly_wino", c0), lambda: ops.pack_conv_weight_split_wino(ops.polyphase_weights(self.conv.weight, c0)[0]).to(self.conv.weight.device))
                    r = ops.conv3d_gcr_split_wino(src0, a[:, :c0].contiguous(), d[:, :c0].contiguous(), pkw, cout, relu=True, with_stats=with_stats,
                                                  act_inv=act_inv, partial=part)
                    return r if with_stats else (r, None)
                r = ops.conv3d_gcr_split(src0, None, a[:, :c0].contiguous(), d[:, :c0].contiguous(), pk0, cout, relu=True, with_stats=with_stats,
                                         act_inv=act_inv, partial=part)
                return r if with_stats else (r, None)
            if wino:
                r = ops.conv3d_gcr_split_wino(src0, a, d, wwino, cout, relu=True, with_stats=with_stats, act_inv=act_inv, **sp)
                return r if with_stats else (r, None)
            r = ops.conv3d_gcr_split(src0, src1, a, d, wpack, cout, relu=True, with_stats=with_stats, act_inv=act_inv, **sp)
            return r if with_stats else (r, None)
        a, d = ops.groupnorm_affine(st0, st1, self.groupnorm.num_groups, self.groupnorm.eps, gamma, beta)
        if with_stats:
            return ops.conv3d_gcr(src0, src1, a, d, wp, self.conv.out_channels, relu=True, with_stats=True)
        return ops.conv3d_gcr(src0, src1, a, d, wp, self.conv.out_channels, relu=True), None


    def _run_at_rest(self, src0, st0, with_stats, sparse, arith, gamma, beta):
        """the affine-in-weights form of run() for an input that is at rest (0 for the scattered volume, sparse['rest_in'] behind it) almost
        everywhere"""
        B, cout, reach = src0.shape[0], self.conv.out_channels, int(sparse["reach"])
        a, d = ops.groupnorm_affine(st0, None, self.groupnorm.num_groups, self.groupnorm.eps, gamma, beta)
        w = self.conv.weight
        if not w.is_contiguous():
            w = w.contiguous()
        rest = sparse.get("rest_in") if reach > 1 else None
        wino = _wino_ok(arith, src0, cout)
        prep = ops.conv_affine_pack(w, a, d, st0, rest, wino=wino)
        # away from the cells the operand is zero: the output is ReLU(0 * scale + K[interior]) = ReLU(K[63]) -- the next layer's rest value
        sparse["rest_out"] = torch.relu(prep.kbias[:, 63]).contiguous()
        sp = {}
        small_in = sparse.get("small_in")
        if (arith.sparse_first_conv and src0.shape[-1] <= 384 and (cout % 128 == 0 or cout % 64 != 0) and min(src0.shape[1:4]) > 2 * reach
                and (reach == 1 or small_in is not None)
                # (the class constants must come from the kernel -- and the pack -- the real launch takes: a Winograd layer handed a small volume it cannot
                #  take, 5^3 from a layer that ran in the direct form, simply runs dense, as in run().  At reach 2 the small volume holds the previous
                #  layer's face values, the operand is NOT zero there and the two forms round differently: constants from the direct pack would break
                #  "occupancy-aware == dense, bit for bit" for such mixed configurations)
                and (not wino or small_in is None or ops.wino_supported(src0.shape[-1], cout, small_in.shape[1:4]))):
            if small_in is None:
                n = 8 if wino else 5              # the Winograd kernels take whole tiles (4 x 8 x 8 / 8 x 8 x 8)
                small_in = torch.zeros((B, n, n, n, src0.shape[-1]), dtype=torch.float32, device=src0.device)
            # a plain dense launch of the same kernel with the same pack over the small all-at-rest volume
            small_out = ops.conv3d_gcr_split_persample(small_in, prep, relu=True)
            sp = dict(tile_active=ops.grid_tile_flags(sparse["flat"], B, src0.shape[1:4], reach), kconst=_class_constants(small_out, reach, cout), kreach=reach)
            sparse["small_out"] = small_out
        r = ops.conv3d_gcr_split_persample(src0, prep, relu=True, with_stats=with_stats, **sp)
        return r if with_stats else (r, None)


class DoubleConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order="gcr", num_groups=8):
        super().__init__()
        if encoder:
            c1_in, c1_out = in_channels, max(out_channels // 2, in_channels)
            c2_in, c2_out = c1_out, out_channels
        else:
            c1_in, c1_out = in_channels, out_channels
            c2_in, c2_out = out_channels, out_channels
        self.add_module("SingleConv1", SingleConv(c1_in, c1_out, kernel_size, order, num_groups))
        self.add_module("SingleConv2", SingleConv(c2_in, c2_out, kernel_size, order, num_groups))

    def run(self, src0, src1=None, stats0=None, stats1=None, sparse_flat=None, arith=None, info=None, rest0=None):
        """sparse_flat: src0 is gn_grid_scatter's volume (flat cell index of every scattered point): both convolutions run occupancy-aware /
        in the affine-in-weights form; info: a dict that receives 'rest_out' [B][Cout], the value the block's output holds away from the
        cells (when that form ran); rest0: that value for src0 of a decoder block (its skip connection)"""
        sp1 = dict(flat=sparse_flat, reach=1) if sparse_flat is not None else None
        y, st = self.SingleConv1.run(src0, src1, stats0, stats1, sparse=sp1, arith=arith, rest0=rest0)
        sp2 = None
        if sp1 is not None and ("small_out" in sp1 or "rest_out" in sp1):
            sp2 = dict(flat=sparse_flat, reach=2, small_in=sp1.get("small_out"), rest_in=sp1.get("rest_out"))
        r = self.SingleConv2.run(y, None, st, sparse=sp2, arith=arith)
        if info is not None and sp2 is not None and "rest_out" in sp2:
            info["rest_out"] = sp2["rest_out"]
        return r


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, apply_pooling=True, conv_layer_order="gcr", num_groups=8):
        super().__init__()
        self.pooling = nn.MaxPool3d(kernel_size=2) if apply_pooling else None
        self.basic_module = DoubleConv(in_channels, out_channels, encoder=True, order=conv_layer_order, num_groups=num_groups)

    def run(self, x, stats=None, sparse_flat=None, arith=None, info=None):
        if self.pooling is not None:
            sparse_flat = None
            c = x.shape[-1]
            if c <= 256 and 256 % (c // 4) == 0:
                x, stats = ops.maxpool3d_2(x, with_stats=True)
            else:
                x, stats = ops.maxpool3d_2(x), None
        return self.basic_module.run(x, None, stats, sparse_flat=sparse_flat, arith=arith, info=info)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, conv_layer_order="gcr", num_groups=8):
        super().__init__()
        self.basic_module = DoubleConv(in_channels, out_channels, encoder=False, order=conv_layer_order, num_groups=num_groups)

    def run(self, encoder_features, x, stats_skip=None, stats_x=None, arith=None, skip_rest=None):
        # cat((encoder_features, upsample_nearest(x)), dim=channel) is never materialised
        return self.basic_module.run(encoder_features, x, stats_skip, stats_x, arith=arith, rest0=skip_rest)


class FinalConv1x1(PackedModule, nn.Conv3d):
    def _pack(self):
        return pack_wb(self.weight.detach().reshape(self.out_channels, self.in_channels), self.bias)

    def run(self, x):
        wp, b, k = self.packed()
        shp = x.shape
        y = ops.linear(x.reshape(-1, shp[-1]), wp, b, None, None, relu=False, K=k)
        return y.reshape(*shp[:-1], self.out_channels)


class Abstract3DUNet(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid=False, basic_module=DoubleConv, f_maps=64, layer_order="gcr",
                 num_groups=8, num_levels=4, is_segmentation=False, testing=False, **kwargs):
        super().__init__()
        if basic_module is not DoubleConv or is_segmentation:
            raise NotImplementedError("only the DoubleConv regression UNet of the GarmentNets pipeline is implemented")
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        self.f_maps = list(f_maps)
        self.encoders = nn.ModuleList([
            Encoder(in_channels if i == 0 else f_maps[i - 1], f, apply_pooling=i > 0, conv_layer_order=layer_order, num_groups=num_groups)
            for i, f in enumerate(f_maps)])
        rf = list(reversed(f_maps))
        self.decoders = nn.ModuleList([
            Decoder(rf[i] + rf[i + 1], rf[i + 1], conv_layer_order=layer_order, num_groups=num_groups) for i in range(len(rf) - 1)])
        self.final_conv = FinalConv1x1(f_maps[0], out_channels, 1)
        self.final_activation = None
        self.arith = None        # arithmetic of forward(x) (None: arith.DEFAULT); run() takes it per call

    def run(self, x, stats=None, pre_final=False, return_stats=False, sparse_flat=None, arith=None):
        """channel-last in, channel-last out (pre_final: stop before the final 1x1x1 convolution -- it is linear, so the decoders can
        fold it into their first layer and sample the f_maps[0]-channel volume instead: networks/conv_implicit_wnf.py UNetResult).  Every kernel that produces a tensor also emits the per-channel statistics the
        next GroupNorm needs (conv / max-pool epilogues), so no activation is re-read for normalisation."""
        feats = []
        for i, enc in enumerate(self.encoders):
            info = {}
            x, stats = enc.run(x, stats, sparse_flat=sparse_flat if i == 0 else None, arith=arith, info=info)
            feats.insert(0, (x, stats, info.get("rest_out")))      # rest_out: encoder 0 behind a scattered volume (affine-in-weights form)
        for dec, (skip, skip_stats, skip_rest) in zip(self.decoders, feats[1:]):
            x, stats = dec.run(skip, x, skip_stats, stats, arith=arith, skip_rest=skip_rest)
        if pre_final:       # return_stats: + (sum, sumsq, V) of the pre-final volume (the decoders derive their input scale from it)
            return (x, stats) if return_stats else x
        return self.final_conv.run(x)

    def forward(self, x):
        """x: (B, C, D, H, W) as in the reference; returns (B, C', D, H, W) (a view over channel-last storage)."""
        stats = getattr(x, "_gn_stats", None)
        return self.run(to_channel_last(x), stats, arith=self.arith).permute(0, 4, 1, 2, 3)


def to_channel_last(x):
    """(B,C,D,H,W) tensor (any strides) -> contiguous [B][D][H][W][C]; free when x already is a channel-last view."""
    return x.permute(0, 2, 3, 4, 1).contiguous()
