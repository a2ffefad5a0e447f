"""ConvImplicitWNFPipeline (inference) -- API twin of /root/reference/networks/conv_implicit_wnf.py:23-338.

Keeps the reference's class names, constructor kwargs, sub-module names (= checkpoint schema: ``pointnet2_nocs``,
``volume_agg.local_nn``, ``unet_3d.abstract_3d_unet``, ``volume_decoder.mlp`` ...), stage methods
(``pointnet2_forward``, ``unet3d_forward``, ``volume_decoder_forward``, ``surface_decoder_forward``,
``mc_surface_decoder_forward``, ``forward``) and result-dict keys, so that predict.py is a drop-in.  Training code
(:340-452) is out of scope.  Feature volumes are stored channel-last; the (B,C,D,H,W) tensors handed out are views.
"""
import os
import threading
import weakref

import torch
from torch import nn

from .. import arith as AR
from .. import ops
from ..batch import Batch
from ..components.mlp import MLP, PackedModule, fold_batchnorm, param_cache
from ..components.unet3d import Abstract3DUNet, DoubleConv, to_channel_last
from ..components.pointnet2 import Segments
from .pointnet2_nocs import PointNet2NOCS


class VolumeFeatureAggregator(nn.Module):
    """per-point MLP -> scatter (max | mean) into a (B,C,G,G,G) volume -- conv_implicit_wnf.py:23-100."""

    def __init__(self, nn_channels=[1024, 1024, 128], batch_norm=True, lower_corner=(0, 0, 0), upper_corner=(1, 1, 1),
                 grid_shape=(32, 32, 32), reduce_method="mean", include_point_feature=True, include_confidence_feature=False):
        super().__init__()
        self.local_nn = MLP(nn_channels, batch_norm=batch_norm)
        self.lower_corner = tuple(lower_corner)
        self.upper_corner = tuple(upper_corner)
        self.grid_shape = tuple(grid_shape)
        self.reduce_method = reduce_method
        self.include_point_feature = include_point_feature
        self.include_confidence_feature = include_confidence_feature

    def prefetch_zero(self, B, device):
        """zero-fill the (B, G, G, G, C) volume of the NEXT forward() on a side stream, now: the fill (17 GB at batch 16, 128^3: 3 ms of
        pure HBM writes) then runs next to PointNet++'s serial farthest-point sampling (16 workgroups on 256 CUs) instead of after it.
        Only callers that WILL run forward() next ask for it (ConvImplicitWNFPipeline.forward, predict._dense_phase) and they call
        drop_prefetch() on their way out, so the buffer never outlives the step that asked for it."""
        if not PREFETCH_ZERO or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        C = self.local_nn[-1][0].out_features if self.local_nn is not None else None
        if C is None:
            return
        dev = _normalise_device(device)
        state = _THREAD_STATE.of(self)
        pre = state.get("prezero")
        if pre is not None and pre[0] == B and pre[1].device == dev:
            return                                   # an earlier prefetch of this shape was never consumed: still zero
        if B * int(torch.tensor(self.grid_shape).prod()) * C * 4 < (64 << 20):
            return                                   # small volumes: the in-stream memset costs microseconds
        main = torch.cuda.current_stream(dev)
        side = state.get("side")
        if side is None or side.device != dev:
            side = state["side"] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)                       # the blocks the allocator hands out were last used on the main stream
        with torch.cuda.stream(side):
            vol, cnt = ops.zeroed_volume(B, self.grid_shape, C, dev)
            ev = torch.cuda.Event()
            ev.record(side)
        state["prezero"] = (B, vol, cnt, ev)

    def drop_prefetch(self):
        """release an unconsumed prefetch_zero buffer (17 GB at batch 16, 128^3)"""
        pre = _THREAD_STATE.of(self).pop("prezero", None)
        if pre is not None:
            pre[1].record_stream(torch.cuda.current_stream(pre[1].device))     # filled on the side stream, freed from this one

    def forward(self, nocs_data):
        B = nocs_data.num_graphs
        conf = nocs_data.pred_confidence
        feats, flat = ops.grid_features(nocs_data.x, nocs_data.pos.contiguous(), nocs_data.sim_points.float().contiguous(),
                                        conf.contiguous(), nocs_data.batch, self.lower_corner, self.upper_corner, self.grid_shape,
                                        self.include_point_feature, self.include_confidence_feature)
        if self.local_nn is not None:
            feats = self.local_nn(feats)
        pre = _THREAD_STATE.of(self).pop("prezero", None)
        prezeroed = None
        if pre is not None and torch.cuda.is_current_stream_capturing():
            pre = None                                # a captured graph must own its memset: never bake "already zero" into it
        if pre is not None and pre[0] == B and pre[1].shape[-1] == feats.shape[1] and pre[1].device == feats.device:
            cur = torch.cuda.current_stream(feats.device)
            cur.wait_event(pre[3])
            pre[1].record_stream(cur)                 # filled on the side stream, consumed (and outlived) by work on this one: the allocator
            pre[2].record_stream(cur)                 # must not hand the block to the side stream's NEXT fill while that work is in flight
            prezeroed = (pre[1], pre[2])
        vol, stats = ops.grid_scatter(feats, flat, B, self.grid_shape, self.reduce_method, with_stats=True, prezeroed=prezeroed)   # [B][G][G][G][C]
        out = vol.permute(0, 4, 1, 2, 3)
        out._gn_stats = stats      # GroupNorm statistics of the (mostly empty) volume, from its occupied cells only
        out._gn_flat = flat        # the occupied cells: the first UNet convolution only visits the tiles that can see one
        return out


class _PerThreadState:
    """per (host thread, module instance) scratch state: the prefetched zero volume and its side stream belong to the thread that asked
    for them -- two threads sharing one model never consume or drop each other's buffer (no shared mutable state on the
    call path).  Entries die with the module (weak keys) or the thread."""

    def __init__(self):
        self._tls = threading.local()

    def of(self, module):
        table = self._tls.__dict__.setdefault("table", weakref.WeakKeyDictionary())
        st = table.get(module)
        if st is None:
            st = table[module] = {}
        return st


_THREAD_STATE = _PerThreadState()


def _normalise_device(device):
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class UNet3D(nn.Module):
    def __init__(self, in_channels, out_channels, f_maps=64, layer_order="gcr", num_groups=8, num_levels=4):
        super().__init__()
        self.abstract_3d_unet = Abstract3DUNet(in_channels=in_channels, out_channels=out_channels, final_sigmoid=False,
                                               basic_module=DoubleConv, f_maps=f_maps, layer_order=layer_order,
                                               num_groups=num_groups, num_levels=num_levels, is_segmentation=False)

    def forward(self, data):
        return self.abstract_3d_unet(data)


class ImplicitWNFDecoder(PackedModule):
    """trilinear feature sampling (border, align_corners) -> MLP -- conv_implicit_wnf.py:120-149.

    The MLP runs as ONE kernel (gn_implicit_decode: hidden activations stay in LDS) when it has the shipped shape
    [C0, N1, N2, out<=4] with N1, N2 multiples of 256; otherwise gn_linear per layer.  Sampling is a separate
    high-occupancy kernel feeding it through a chunk buffer (2^20 rows x 32 channels = 134 MB: half a 128^3 lattice per launch pair; 2^18
    rows measured 1 ms slower per 16-garment step, the whole lattice no faster) small enough to stay in the 256 MB Infinity Cache
    (measured on MI355X, 128^3 lattice: sample 1.96 ms + MLP 4.47 ms vs 9.55 ms for sampling inside the MLP kernel,
    where the latency-bound gathers cannot overlap the matrix-core phases of the only two resident workgroups)."""

    ROWS_PER_CHUNK = int(os.environ.get("GARMENTNETS_DECODE_CHUNK", 1 << 20))
    fused = True

    def __init__(self, nn_channels=(128, 512, 512, 1), batch_norm=True):
        super().__init__()
        self.mlp = MLP(list(nn_channels), batch_norm=batch_norm)
        self.nn_channels = tuple(nn_channels)
        self.out_channels = nn_channels[-1]
        self.arith = None          # arithmetic of a direct call (None: arith.DEFAULT); the pipeline passes its own per call

    def _pack(self):
        ch = self.nn_channels
        if not (len(ch) == 4 and ch[0] % 32 == 0 and ch[1] % 256 == 0 and ch[2] % 256 == 0 and ch[3] <= 4):
            return None
        layers, raw = [], []
        for i, block in enumerate(self.mlp):
            w = block[0].weight.detach().float()
            b = block[0].bias.detach().float().contiguous()
            sc, sh = fold_batchnorm(block[2]) if len(block) > 2 else (None, None)
            layers.append((ops.pack_kpair(w) if i < 2 else w.contiguous(), b, sc, sh, ch[i + 1]))
            raw.append((w, b, sc, sh))
        # the shipped [128,256,256,out] decoder also gets the split-operand pack (csrc/decode_split.hip, Arith.decode_mode)
        split = ops.pack_decode_split(raw).to(w.device) if tuple(ch[:3]) == (128, 256, 256) else None
        return tuple(layers) + (split,)

    def folded_pack(self, final_conv):
        """The same decoder with the UNet's final 1x1x1 convolution (linear, no activation) folded into its first layer:
        W1' = W1 Wf, b1' = b1 + W1 bf (fp64 on the host).  It runs on rows sampled from the PRE-final feature volume (f_maps[0] = 32
        channels instead of 128): trilinear interpolation is linear and its weights sum to one, so sample(Wf x + bf) = Wf sample(x) +
        bf up to rounding.  The 128-channel volume is then never written or read (17 GB per 16-garment batch at 128^3), the
        sampler moves a quarter of the bytes and layer 1 does a quarter of the FLOPs.  -> packed() layout, or None when not foldable."""
        ch = self.nn_channels
        if not (self.fused and len(ch) == 4 and ch[0] == final_conv.out_channels and final_conv.in_channels % 32 == 0 and ch[1] % 256 == 0
                and ch[2] % 256 == 0 and ch[3] <= 4 and final_conv.kernel_size == (1, 1, 1)):
            return None
        key = (id(final_conv), final_conv.weight._version, final_conv.bias._version, final_conv.weight.device) + tuple(p._version for p in self.parameters())

        def build():
            wf = final_conv.weight.detach().double().reshape(final_conv.out_channels, final_conv.in_channels)
            bf = final_conv.bias.detach().double()
            layers, raw = [], []
            for i, block in enumerate(self.mlp):
                w = block[0].weight.detach().double()
                b = block[0].bias.detach().double()
                if i == 0:
                    b = b + w @ bf
                    w = w @ wf
                w, b = w.float(), b.float().contiguous()
                sc, sh = fold_batchnorm(block[2]) if len(block) > 2 else (None, None)
                layers.append((ops.pack_kpair(w) if i < 2 else w.contiguous(), b, sc, sh, ch[i + 1]))
                raw.append((w, b, sc, sh))
            # the split-operand pack: the shipped hidden width 256 and the class default 512 (conv_implicit_wnf.py:122), csrc/decode_split.hip
            split = ops.pack_decode_split(raw).to(w.device) if (final_conv.in_channels, ch[1], ch[2]) in ((32, 256, 256), (32, 512, 512)) else None
            return tuple(layers) + (split,)
        return param_cache(self, "_folded").get(key, "folded", build)

    def _decode_rows(self, vol_b, out, query=None, Q=0, layers=None, xscale=None, arith=None):
        """vol_b [D][H][W][C]: the decoder's own input volume, or (with the folded `layers`) the UNet's pre-final volume;
        xscale: this garment's (s, 1/s) input scale for the split-operand kernel (ops.decoder_input_scale)"""
        arith = arith or AR.DEFAULT
        M = out.shape[0]
        if layers is None:
            layers = self.packed() if self.fused else None
        split = layers is not None and layers[3] is not None and arith.decode_mode == "f16x2"
        if query is None and split and arith.fused_lattice and ops.lattice_split_supported(vol_b, layers[3]):
            # the lattice sampler inside the decoder kernel -- one launch per garment, no sampled-row buffer
            ops.implicit_decode_lattice_split(vol_b, Q, layers[3], out, xscale=xscale)
            if xscale is not None:          # garments the device marked unsafe for fp16 planes: the gated fp32 twin samples for itself
                ops.implicit_decode(vol_b, layers[:3], Q=Q, m0=0, M=M, out=out, run_if=xscale[2:3])
            return
        chunk = self.ROWS_PER_CHUNK
        if query is None:                           # lattice: whole i-slabs per chunk (the brick sampler's unit)
            chunk = max(1, chunk // (Q * Q)) * Q * Q
        buf = ops.new_rows(min(M, chunk), vol_b.shape[-1], vol_b.device)
        for m0 in range(0, M, chunk):
            m = min(chunk, M - m0)
            if query is not None:
                s = ops.trilinear_sample(vol_b, query=query[m0:m0 + m], out=buf[:m])
            else:
                s = ops.trilinear_sample(vol_b, Q=Q, m0=m0, M=m, out=buf[:m])
            if split:
                ops.implicit_decode_split(s, layers[3], out=out[m0:m0 + m], xscale=xscale)
                if xscale is not None:      # the garments the device marked unsafe for fp16 planes: gated fp32 twin (a no-op otherwise)
                    ops.implicit_decode(None, layers[:3], M=m, out=out[m0:m0 + m], xin=s, run_if=xscale[2:3])
            elif layers is not None:
                ops.implicit_decode(None, layers[:3], M=m, out=out[m0:m0 + m], xin=s)
            else:
                out[m0:m0 + m] = self.mlp(s)

    # queries of a whole batch in one set of launches when their sampled rows fit the cache they are handed over in (the surface decoders of
    # predict.py:184-187: 16 x ~49 000 vertices x 32 channels = 100 MB); larger query sets (and the lattice) go garment by garment in 1 M-row chunks
    BATCH_ROWS_BYTES = int(os.environ.get("GARMENTNETS_DECODE_BATCH_BYTES", 192 << 20))

    def _decode_queries_batched(self, vol, q, out, layers, xs, arith):
        """vol (B,D,H,W,C) channel-last, q (B,M,3), out (B,M,OUT) through the split-operand decoder in three launches for the whole batch (sampler, decoder,
        gated fp32 twin); -> False when this path does not apply (the caller loops over the garments: same results row for row)"""
        B, M = q.shape[:2]
        C = vol.shape[-1]
        if (B < 2 or M == 0 or layers is None or layers[3] is None or arith.decode_mode != "f16x2" or not vol.is_contiguous()
                or B * M * ops.pad4(C) * 4 > self.BATCH_ROWS_BYTES or out.stride(0) != M * out.stride(1)):
            return False
        rows = ops.trilinear_sample_batch(vol, q)
        ops.implicit_decode_split_batch(rows, layers[3], out, xscale=xs)
        if xs is not None:                  # the garments the device marked unsafe for fp16 planes: gated fp32 twin (a no-op otherwise)
            ops.implicit_decode_batch(rows, layers[:3], out, run_if=xs[:, 2:], run_if_stride=xs.stride(0))
        return True

    def forward(self, features_grid, query_points, arith=None):
        """features_grid (B,C,D,H,W), query_points (B,M,3) in [0,1] -> (B,M,out)"""
        arith = arith or self.arith or AR.DEFAULT
        vol = to_channel_last(features_grid)
        B, M = query_points.shape[:2]
        q = query_points.float().contiguous()
        out = torch.empty((B, M, self.out_channels), dtype=torch.float32, device=vol.device)
        xs = self._volume_scales(features_grid, vol, arith)
        if self._decode_queries_batched(vol, q, out, self.packed() if self.fused else None, xs, arith):
            return out
        for b in range(B):
            self._decode_rows(vol[b], out[b], query=q[b], xscale=None if xs is None else xs[b], arith=arith)
        return out

    def _volume_scales(self, owner, vol, arith):
        """(B, 4) input scales of the split-operand decoder kernel for the channel-last volume the rows are sampled from, or None when
        that kernel is not the one that runs.  One gn_channel_stats pass over the volume, remembered ON the caller's tensor object
        `owner` together with its version counter (the reference's chunk loop calls the decoder 8 times per garment on one volume): the
        entry dies with the tensor, so a recycled allocation can never be mistaken for it."""
        layers = self.packed() if self.fused else None
        if layers is None or layers[3] is None or arith.decode_mode != "f16x2":
            return None
        cached = getattr(owner, "_gn_chan_stats", None)
        if cached is None or cached[0] != owner._version:
            cached = (owner._version, ops.channel_stats(vol))
            try:
                owner._gn_chan_stats = cached
            except AttributeError:
                pass
        stats = cached[1]
        return ops.decoder_input_scale(stats[1], stats[2], layers[3].smax)

    def decode_lattice(self, features_grid, Q, arith=None):
        """All (Q,Q,Q) lattice queries of predict.py:145-157 without materialising them -> (B,Q,Q,Q[,out])"""
        arith = arith or self.arith or AR.DEFAULT
        vol = to_channel_last(features_grid)
        B = vol.shape[0]
        out = torch.empty((B, Q * Q * Q, self.out_channels), dtype=torch.float32, device=vol.device)
        xs = self._volume_scales(features_grid, vol, arith)
        for b in range(B):
            self._decode_rows(vol[b], out[b], Q=Q, xscale=None if xs is None else xs[b], arith=arith)
        return out.reshape(B, Q, Q, Q, self.out_channels).squeeze(-1)

    def run_on(self, unet3d_result, query_points=None, Q=0, arith=None):
        """decode against a unet3d_forward result: through the folded first layer when the result still carries the pre-final
        volume (UNetResult) and this decoder can absorb the final convolution, else on the materialised 128-channel volume.
        query_points (B,M,3) -> (B,M,out); query_points None -> the (Q,Q,Q) lattice -> (B,Q,Q,Q[,out])"""
        arith = arith or self.arith or AR.DEFAULT
        layers = None
        if isinstance(unet3d_result, UNetResult) and arith.fold_final_conv:
            layers = self.folded_pack(unet3d_result.final_conv)
        if layers is None:
            vol = unet3d_result["out_feature_volume"]
            return self.decode_lattice(vol, Q, arith) if query_points is None else self(vol, query_points, arith)
        vol = unet3d_result.pre_final
        B = vol.shape[0]
        xs = None
        if layers[3] is not None and arith.decode_mode == "f16x2":
            xs = unet3d_result.input_scales(layers[3].smax)
        if query_points is None:
            out = torch.empty((B, Q * Q * Q, self.out_channels), dtype=torch.float32, device=vol.device)
            for b in range(B):
                self._decode_rows(vol[b], out[b], Q=Q, layers=layers, xscale=None if xs is None else xs[b], arith=arith)
            return out.reshape(B, Q, Q, Q, self.out_channels).squeeze(-1)
        q = query_points.float().contiguous()
        out = torch.empty((B, q.shape[1], self.out_channels), dtype=torch.float32, device=vol.device)
        if self._decode_queries_batched(vol, q, out, layers, xs, arith):
            return out
        for b in range(B):
            self._decode_rows(vol[b], out[b], query=q[b], layers=layers, xscale=None if xs is None else xs[b], arith=arith)
        return out


# zero-fill of the scattered volume overlapped with PointNet++ (VolumeFeatureAggregator.prefetch_zero)
PREFETCH_ZERO = os.environ.get("GARMENTNETS_PREFETCH_ZERO", "1") != "0"

class UNetResult(dict):
    """unet3d_forward's result: {'out_feature_volume': (B,128,D,H,W)} as in the reference, except that the 128-channel volume is
    only materialised (one gn_linear over all voxels) if somebody actually reads it; the decoders do not (see folded_pack)."""

    def __init__(self, pre_final, final_conv, pre_stats=None):
        super().__init__()
        self.pre_final, self.final_conv = pre_final, final_conv          # [B][D][H][W][f_maps[0]] channel-last
        self.pre_stats = pre_stats                                       # (sum, sumsq, V) of pre_final from the last conv's epilogue, or None
        self._scales = {}                                                # smax -> (B, 2) decoder input scales

    def input_scales(self, smax):
        """(B, 2) run-time input scales (s, 1/s) of the split-operand decoder kernel for this volume (ops.decoder_input_scale)"""
        if smax not in self._scales:
            if self.pre_stats is None:
                self.pre_stats = ops.channel_stats(self.pre_final)
            self._scales[smax] = ops.decoder_input_scale(self.pre_stats[1], self.pre_stats[2], smax)
        return self._scales[smax]

    def select(self, b0, b1):
        """the same result for garments b0..b1-1 (predict.py slices out_feature_volume[[i]] per sample)"""
        st = None if self.pre_stats is None else (self.pre_stats[0][b0:b1], self.pre_stats[1][b0:b1], self.pre_stats[2])
        sub = UNetResult(self.pre_final[b0:b1], self.final_conv, st)
        sub._scales = {k: v[b0:b1] for k, v in self._scales.items()}
        if dict.__contains__(self, "out_feature_volume"):
            dict.__setitem__(sub, "out_feature_volume", dict.__getitem__(self, "out_feature_volume")[b0:b1])
        return sub

    def _materialise(self):
        if not dict.__contains__(self, "out_feature_volume"):
            dict.__setitem__(self, "out_feature_volume", self.final_conv.run(self.pre_final).permute(0, 4, 1, 2, 3))

    def __getitem__(self, k):
        if k == "out_feature_volume":
            self._materialise()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "out_feature_volume" or dict.__contains__(self, k)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def get(self, k, default=None):
        return self[k] if k in self else default

    # the rest of the Mapping surface sees the reference's one-key dict too (CPython's dict(u3) / copy / len fast paths bypass keys())
    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return 1

    def copy(self):
        self._materialise()
        return dict(dict.items(self))

    def __reduce__(self):
        self._materialise()
        return (dict, (dict(dict.items(self)),))

    def __repr__(self):
        return "UNetResult(out_feature_volume=%s)" % ("<lazy>" if not dict.__contains__(self, "out_feature_volume") else tuple(dict.__getitem__(self, "out_feature_volume").shape),)


class ConvImplicitWNFPipeline(nn.Module):
    def __init__(self, pointnet2_params, volume_agg_params, unet3d_params, volume_decoder_params, surface_decoder_params,
                 mc_surface_decoder_params=None, learning_rate=1e-4, loss_type="l2", volume_loss_weight=1.0,
                 surface_loss_weight=1.0, mc_surface_loss_weight=0, volume_classification=False, volume_task_space=False,
                 vis_per_items=0, max_vis_per_epoch_train=0, max_vis_per_epoch_val=0, batch_size=None):
        super().__init__()
        self.hparams = dict(pointnet2_params=dict(pointnet2_params), volume_agg_params=dict(volume_agg_params),
                            unet3d_params=dict(unet3d_params), volume_decoder_params=dict(volume_decoder_params),
                            surface_decoder_params=dict(surface_decoder_params),
                            mc_surface_decoder_params=None if mc_surface_decoder_params is None else dict(mc_surface_decoder_params),
                            mc_surface_loss_weight=mc_surface_loss_weight, volume_task_space=volume_task_space)
        self.pointnet2_nocs = PointNet2NOCS(**pointnet2_params)
        self.volume_agg = VolumeFeatureAggregator(**volume_agg_params)
        self.unet_3d = UNet3D(**unet3d_params)
        self.volume_decoder = ImplicitWNFDecoder(**volume_decoder_params)
        self.surface_decoder = ImplicitWNFDecoder(**surface_decoder_params)
        self.mc_surface_decoder = None
        if mc_surface_loss_weight > 0:
            self.mc_surface_decoder = ImplicitWNFDecoder(**mc_surface_decoder_params)
        self.volume_task_space = volume_task_space
        self.batch_size = batch_size
        self.arith = AR.DEFAULT      # this model's arithmetic (an immutable arith.Arith); every stage method also takes arith= per call

    # -- checkpoint ------------------------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
        """Reads a Lightning-style checkpoint {'state_dict', 'hyper_parameters'} (predict.py:101) without Lightning."""
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        hp = dict(ckpt["hyper_parameters"])
        hp.update(overrides)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"])
        return model

    def save_checkpoint(self, path):
        torch.save({"state_dict": self.state_dict(), "hyper_parameters": self.hparams}, path)

    @property
    def device(self):
        return self.pointnet2_nocs.device

    # -- stages ----------------------------------------------------------------------------------------------
    def pointnet2_forward(self, data, prefetch_volume=False):
        """prefetch_volume: the caller runs unet3d_forward next (and volume_agg.drop_prefetch() on its way out): start the zero-fill
        of the feature volume beside PointNet++ (VolumeFeatureAggregator.prefetch_zero)"""
        sizes = data._sizes if hasattr(data, "_sizes") else None
        if prefetch_volume and sizes is not None and data.pos.is_cuda:   # (no host sizes: the batch size would cost a device synchronisation here)
            self.volume_agg.prefetch_zero(len(sizes), data.pos.device)
        seg = Segments.of(data.batch, sizes if sizes is not None else getattr(data, "sizes", None))     # the batch's sizes travel with THIS call
        result = self.pointnet2_nocs(data, seg=seg)
        bins = self.pointnet2_nocs.nocs_bins
        _, confidence, pred_nocs = ops.nocs_head(result["per_point_logits"], bins)
        result["nocs_data"] = Batch(sizes=seg.sizes, x=result["per_point_features"], pos=pred_nocs, batch=result["per_point_batch_idx"],
                                    sim_points=data.pos, pred_confidence=confidence)
        return result

    def unet3d_forward(self, pointnet2_result, arith=None):
        arith = arith or self.arith
        in_feature_volume = self.volume_agg(pointnet2_result["nocs_data"])
        net = self.unet_3d.abstract_3d_unet
        pre, st = net.run(to_channel_last(in_feature_volume), getattr(in_feature_volume, "_gn_stats", None), pre_final=True, return_stats=True,
                          sparse_flat=getattr(in_feature_volume, "_gn_flat", None), arith=arith)
        return UNetResult(pre, net.final_conv, st)   # ['out_feature_volume'] materialises the reference's tensor on demand

    def volume_decoder_forward(self, unet3d_result, query_points, arith=None):
        out = self.volume_decoder.run_on(unet3d_result, query_points, arith=arith or self.arith)
        return {"out_features": out, "pred_volume_value": out.view(*out.shape[:-1])}

    def surface_decoder_forward(self, unet3d_result, query_points, arith=None):
        return {"out_features": self.surface_decoder.run_on(unet3d_result, query_points, arith=arith or self.arith)}

    def mc_surface_decoder_forward(self, unet3d_result, query_points, arith=None):
        return {"out_features": self.mc_surface_decoder.run_on(unet3d_result, query_points, arith=arith or self.arith)}

    def volume_lattice_forward(self, unet3d_result, volume_size, arith=None):
        """Whole (Q,Q,Q) WNF volume per garment in one pass (replaces the 64^3 chunk loop of predict.py:145-157)."""
        return {"pred_volume": self.volume_decoder.run_on(unet3d_result, None, Q=volume_size, arith=arith or self.arith)}

    @staticmethod
    def get_aabb_scale_offset(aabb, padding=0.05):
        """conv_implicit_wnf.py:299-313: per-sample scale / offset that maps the task-space bounding box (B,2,3) into the unit NOCS cube
        (x, y centred on 0.5, the top of z at 1 - padding)"""
        nocs_radius = 0.5 - padding
        radius = torch.max(torch.abs(aabb), dim=1)[0][:, :2]
        radius_scale = torch.min(nocs_radius / radius, dim=1)[0]
        z_scale = (nocs_radius * 2) / (aabb[:, 1, 2] - aabb[:, 0, 2])
        scale = torch.minimum(radius_scale, z_scale)
        offset = torch.full((len(aabb), 3), 0.5, dtype=aabb.dtype, device=aabb.device)
        offset[:, 2] = 1 - padding - aabb[:, 1, 2] * scale
        return scale, offset

    def apply_volume_task_space(self, data, pointnet2_result):
        """conv_implicit_wnf.py:279-297: the gridding runs on the normalised SIMULATION coordinates of the points instead of their
        predicted NOCS coordinates (first sample's scale / offset for the whole batch, as the reference)"""
        scale, offset = self.get_aabb_scale_offset(data.cloth_sim_aabb.to(data.pos.device))
        nd = pointnet2_result["nocs_data"]
        new_nd = Batch(sizes=nd._sizes, **{k: getattr(nd, k) for k in nd.keys})
        new_nd.pos = ((data.pos * scale[0]) + offset[0]).to(torch.float32).contiguous()
        out = dict(pointnet2_result)
        out["nocs_data"] = new_nd
        return out

    def forward(self, data, arith=None):
        """conv_implicit_wnf.py:315-338"""
        arith = arith or self.arith
        try:
            p2 = self.pointnet2_forward(data, prefetch_volume=True)
            if self.volume_task_space:
                p2 = self.apply_volume_task_space(data, p2)
            u3 = self.unet3d_forward(p2, arith)
        finally:
            self.volume_agg.drop_prefetch()
        result = {
            "pointnet2_result": p2,
            "unet3d_result": u3,
            "volume_decoder_result": self.volume_decoder_forward(u3, data.volume_query_points, arith),
            "surface_decoder_result": self.surface_decoder_forward(u3, data.surf_query_points, arith),
        }
        if self.mc_surface_decoder is not None:
            result["mc_surface_decoder_result"] = self.mc_surface_decoder_forward(u3, data.mc_surf_query_points, arith)
        return result
