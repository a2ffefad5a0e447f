"""Thin torch-tensor wrappers over the C ABI (include/garmentnets_hip.h).

torch is used for device memory and streams only; every function below launches hand-written HIP kernels on the
current torch stream.  Tensors must live on a ROCm device -- there is no CPU path.
"""
import ctypes
import math
import threading

import numpy as np
import torch

from . import _lib

_i32 = torch.int32


class _CallState(threading.local):
    """per HOST THREAD: the device of the tensors of the C-ABI call being assembled (arguments are evaluated left to right, _stream()
    last).  Thread-local, so two threads driving two models / devices never see each other's half-assembled call"""
    dev = None


_CALL = _CallState()


def _stream():
    """torch's current stream ON THE DEVICE OF THE CALL'S TENSORS (not of torch's current device: a model on cuda:1 with the
    process default device 0 must launch on cuda:1's stream; the C ABI binds the HIP device to the stream's, csrc/common.h gn_stream)"""
    dev, _CALL.dev = _CALL.dev, None
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)            # the null stream means "current device" to HIP: make the two agree
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.GarmentNetsHipError("garmentnets_amd ops need tensors on the GPU (no CPU fallback)")
    if _CALL.dev is None:
        _CALL.dev = t.device
    elif _CALL.dev != t.device:
        dev, _CALL.dev = _CALL.dev, None
        raise _lib.GarmentNetsHipError(f"garmentnets_amd ops: tensors on different devices ({dev} vs {t.device})")
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def pad4(c):
    return (c + 3) // 4 * 4


def rows_view(t):
    """(rows, ld) of a 2-D tensor whose rows are contiguous (stride(1)==1); ld = stride(0)."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), "need row-major rows"
    return t.shape[0], (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def new_rows(n, c, device):
    """[n][c] fp32 buffer whose leading dimension is padded to a multiple of 4 (16-byte rows for the GEMM loader)."""
    buf = torch.empty((n, pad4(c)), dtype=torch.float32, device=device)
    return buf[:, :c]


# ------------------------------------------------------------------------------------------------ points
def fps_count(n, ratio):
    """torch_cluster: ceil(n * ratio) evaluated in float32."""
    return int(math.ceil(float(np.float32(n) * np.float32(ratio))))


def segment_ptr(batch, B):
    _chk(batch, torch.int64, "batch")
    ptr = torch.empty(B + 1, dtype=_i32, device=batch.device)
    _lib.call("gn_segment_ptr", _p(batch), batch.numel(), B, _p(ptr), _stream())
    return ptr


def fps(pos, ptr, out_ptr, max_points, m_total, start_idx=None, gap_out=None, nested_gap=None):
    """start_idx: optional int32 [B] local start point per example (None = first point).
    gap_out / nested_gap: float32 [B] (include/garmentnets_hip.h gn_fps_nested): a cascade's second level hands in the first level's gap_out and
    gets the prefix 0..m-1 of every example whose first-level running maximum stayed positive -- the same indices without the serial steps."""
    _chk(pos, torch.float32, "pos")
    idx = torch.empty(m_total, dtype=_i32, device=pos.device)
    if gap_out is None and nested_gap is None:
        _lib.call("gn_fps", _p(pos), _p(ptr), _p(out_ptr), _p(start_idx), ptr.numel() - 1, int(max_points), _p(idx), _stream())
    else:
        if gap_out is not None:
            _chk(gap_out, torch.float32, "gap_out")
        if nested_gap is not None:
            _chk(nested_gap, torch.float32, "nested_gap")
        _lib.call("gn_fps_nested", _p(pos), _p(ptr), _p(out_ptr), _p(start_idx), ptr.numel() - 1, int(max_points), _p(idx), _p(gap_out), _p(nested_gap),
                  _stream())
    return idx


def ball_query(pos, ptr, centre_idx, centre_ptr, r, K=64):
    M = centre_idx.numel()
    nbr = torch.empty((M, K), dtype=_i32, device=pos.device)
    cnt = torch.empty(M, dtype=_i32, device=pos.device)
    r2 = float(np.float32(float(r) * float(r)))
    _lib.call("gn_ball_query", _p(pos), _p(ptr), _p(centre_idx), _p(centre_ptr), ptr.numel() - 1, M, r2, K, _p(nbr), _p(cnt), _stream())
    return nbr, cnt


def sa_gather(x, pos, centre_idx, nbr, self_loops=True, self_src=None):
    """self_src: int32 [M] scope of the self-loop rule (include/garmentnets_hip.h: gn_sa_gather_scoped) or None = PyG's literal rule"""
    M, K = nbr.shape
    C = 0 if x is None else x.shape[1]
    S = K + (1 if self_loops else 0)
    out = new_rows(M * S, C + 3, pos.device)
    slot_src = torch.empty(M * S, dtype=_i32, device=pos.device)
    ldx = 0 if x is None else rows_view(x)[1]
    _lib.call("gn_sa_gather_scoped", _p(x), ldx, C, _p(pos), _p(centre_idx), _p(nbr), M, K, 1 if self_loops else 0,
              _p(_chk(self_src, _i32, "self_src") if self_src is not None else None), _p(out), out.stride(0), _p(slot_src), _stream())
    return out, slot_src, S


def segment_max(h, slot_src, M, S):
    C = h.shape[1]
    out = new_rows(M, C, h.device)
    _lib.call("gn_segment_max", _p(h), rows_view(h)[1], _p(slot_src), M, S, C, _p(out), out.stride(0), _stream())
    return out


class SaFusedPack:
    """weights / epilogue tables of gn_sa_fused for one edge MLP [cin+3, n1, n2, n3]"""

    def __init__(self, w1p, w2p, w3p, tab, cin, dims):
        self.w1p, self.w2p, self.w3p, self.tab, self.cin, self.dims = w1p, w2p, w3p, tab, int(cin), tuple(int(v) for v in dims)
        self.cout = self.dims[2]
        self.macs_per_edge = (self.cin + 3) * self.dims[0] + self.dims[0] * self.dims[1] + self.dims[1] * self.dims[2]

    def to(self, device):
        return SaFusedPack(self.w1p.to(device), self.w2p.to(device), self.w3p.to(device), self.tab.to(device), self.cin, self.dims)


def sa_fused_supported(cin, dims):
    return len(dims) == 3 and bool(_lib.load().gn_sa_fused_supported(int(cin), int(dims[0]), int(dims[1]), int(dims[2])))


def pack_sa_fused(layers):
    """layers = ((w1,b1,s1,t1), (w2,b2,s2,t2), (w3,b3,s3,t3)): Linear weight (n, k) / bias and the folded eval-BatchNorm scale / shift (or
    None) of the three blocks of a PointConv local_nn; w1 is (n1, cin + 3) -> SaFusedPack.  A-fragment order of csrc/sa_fused.hip:
    Wp[nb][kb][qq][lane = 32 h + r][i] = W[32 nb + r][32 kb + 8 qq + 4 h + i] (zero beyond k); tables in accumulator-register order:
    register q of lane half h of block nb is unit 32 nb + 8 (q >> 2) + (q & 3) + 4 h."""
    ar = torch.arange
    packs, tabs, dims = [], [], []
    for w, b, sc, sh in layers:
        w = w.detach().float().cpu()
        n, k = w.shape
        assert n % 32 == 0
        kb_n = (((k + 7) // 8 * 8) + 31) // 32
        wz = torch.zeros((n, kb_n * 32), dtype=torch.float32)
        wz[:, :k] = w
        nb, kb, qq, h, r, i = torch.meshgrid(ar(n // 32), ar(kb_n), ar(4), ar(2), ar(32), ar(4), indexing="ij")
        packs.append(wz[32 * nb + r, 32 * kb + 8 * qq + 4 * h + i].contiguous())                 # [nb][kb][qq][h][r][4]
        nbt, ht, q = torch.meshgrid(ar(n // 32), ar(2), ar(16), indexing="ij")
        u = 32 * nbt + 8 * (q >> 2) + (q & 3) + 4 * ht                                            # [nb][2][16]
        one = lambda v, fill: (torch.full((n,), fill) if v is None else v.detach().float().cpu())
        tabs.append(torch.stack((one(b, 0.0)[u], one(sc, 1.0)[u], one(sh, 0.0)[u]), dim=2).reshape(-1))   # [nb][2][3][16]
        dims.append(n)
    cin = layers[0][0].shape[1] - 3
    return SaFusedPack(packs[0], packs[1], packs[2], torch.cat(tabs).contiguous(), cin, dims)


def sa_fused(x, pos, centre_idx, nbr, cnt, pack, self_loops=True, self_src=None):
    """fps centres + ball-query table -> [M][n3] set-abstraction features (PointConv(local_nn, max) in one kernel, csrc/sa_fused.hip)"""
    M, K = nbr.shape
    out = new_rows(M, pack.cout, pos.device)
    ldx = 0 if x is None else rows_view(x)[1]
    _lib.call("gn_sa_fused_scoped", _p(x), ldx, pack.cin, _p(_chk(pos, torch.float32, "pos")), _p(centre_idx), _p(nbr), _p(cnt), M, K, 1 if self_loops else 0,
              _p(_chk(self_src, _i32, "self_src") if self_src is not None else None), _p(pack.w1p), _p(pack.w2p), _p(pack.w3p), _p(pack.tab), pack.dims[0], pack.dims[1], pack.dims[2], _p(out), out.stride(0), _stream())
    return out


def global_max_pool(h, ptr, B):
    C = h.shape[1]
    out = new_rows(B, C, h.device)
    _lib.call("gn_global_max_pool", _p(h), rows_view(h)[1], _p(ptr), B, C, _p(out), out.stride(0), _stream())
    return out


def knn_interpolate(xs, ps, ptr_s, pq, ptr_q, k, out=None):
    Nq, C = pq.shape[0], xs.shape[1]
    if out is None:
        out = new_rows(Nq, C, xs.device)
    _lib.call("gn_knn_interpolate", _p(xs), rows_view(xs)[1], _p(ps), _p(ptr_s), _p(pq), _p(ptr_q), ptr_s.numel() - 1, Nq, C,
              int(k), _p(out), rows_view(out)[1], _stream())
    return out


def linear(x, w, bias=None, bn_scale=None, bn_shift=None, relu=False, out=None, K=None):
    """x [M][K] rows, w [N][ldw] (packed, possibly K-padded) -> [M][N]."""
    M = x.shape[0]
    K = x.shape[1] if K is None else K
    N = w.shape[0]
    if out is None:
        out = new_rows(M, N, x.device)
    _lib.call("gn_linear", _p(x), rows_view(x)[1], _p(w), rows_view(w)[1], _p(bias), _p(bn_scale), _p(bn_shift), 1 if relu else 0,
              M, N, K, _p(out), rows_view(out)[1], _stream())
    return out


def nocs_head(logits, bins):
    N = logits.shape[0]
    dev = logits.device
    idx = torch.empty((N, 3), dtype=torch.int64, device=dev)
    conf = torch.empty((N, 3), dtype=torch.float32, device=dev)
    nocs = torch.empty((N, 3), dtype=torch.float32, device=dev)
    _lib.call("gn_nocs_head", _p(logits), rows_view(logits)[1], N, bins, _p(idx), _p(conf), _p(nocs), _stream())
    return idx, conf, nocs


# ------------------------------------------------------------------------------------------------ gridding
def _f3(v):
    return (ctypes.c_float * 3)(*[float(a) for a in v])


def _i3(v):
    return (ctypes.c_int * 3)(*[int(a) for a in v])


def grid_features(feat, nocs, sim_pos, conf, batch, lower, upper, grid_shape, include_point=True, include_conf=True):
    N, Cf = feat.shape
    C = Cf + (6 if include_point else 0) + (3 if include_conf else 0)
    out = new_rows(N, C, feat.device)
    flat = torch.empty(N, dtype=_i32, device=feat.device)
    _lib.call("gn_grid_features", _p(feat), rows_view(feat)[1], Cf, _p(_chk(nocs, torch.float32, "nocs")),
              _p(_chk(sim_pos, torch.float32, "sim_pos")), _p(_chk(conf, torch.float32, "conf")), _p(_chk(batch, torch.int64, "batch")),
              N, _f3(lower), _f3(upper), _i3(grid_shape), 1 if include_point else 0, 1 if include_conf else 0, _p(out),
              out.stride(0), _p(flat), _stream())
    return out, flat


def zeroed_volume(B, grid_shape, C, device):
    """(vol, count workspace) of grid_scatter, zero-filled on torch's CURRENT stream (callers put it on a side stream)"""
    vol = torch.zeros((B,) + tuple(grid_shape) + (C,), dtype=torch.float32, device=device)
    cnt = torch.zeros(B * int(np.prod(grid_shape)), dtype=_i32, device=device)
    return vol, cnt


def grid_scatter(src, flat_idx, B, grid_shape, reduce, with_stats=False, prezeroed=None):
    """-> channel-last volume [B][G0][G1][G2][C] (and its per-channel statistics, from the occupied cells only).
    prezeroed: (vol, cnt) from zeroed_volume that the caller has already ordered before this call"""
    N, C = src.shape
    cps = int(np.prod(grid_shape))
    cells = B * cps
    if prezeroed is not None:
        vol, cnt = prezeroed
        assert vol.shape == (B,) + tuple(grid_shape) + (C,) and cnt.numel() == cells
    else:
        vol = torch.empty((B,) + tuple(grid_shape) + (C,), dtype=torch.float32, device=src.device)
        cnt = torch.empty(cells, dtype=_i32, device=src.device)
    code = {"max": 0, "mean": 1}[reduce]
    nbytes = _lib.load().gn_grid_scatter_workspace_bytes(N, C, code)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=src.device) if nbytes else None
    _lib.call("gn_grid_scatter", _p(src), rows_view(src)[1], _p(flat_idx), N, C, cells, code, _p(vol), _p(cnt), _p(ws), nbytes,
              0 if prezeroed is None else 1, _stream())
    if not with_stats:
        return vol
    s, q = _stats_buffers(B, C, src.device, True)
    _lib.call("gn_grid_stats", _p(vol), _p(flat_idx), N, C, cps, B, _p(cnt), _p(s), _p(q), _stream())
    return vol, (s, q, cps)


# ------------------------------------------------------------------------------------------------ UNet
def channel_stats(x):
    """x channel-last [B][D][H][W][C] -> (sum, sumsq) fp64 [B][C]"""
    B, C = x.shape[0], x.shape[-1]
    V = x.numel() // (B * C)
    s, q = _stats_buffers(B, C, x.device, True)
    _lib.call("gn_channel_stats", _p(x), B, V, C, _p(s), _p(q), _stream())
    return s, q, V


def groupnorm_affine(st0, st1, groups, eps, gamma, beta, with_act_scale=False):
    """-> (a, d) [B][C0+C1]; with_act_scale: -> (a, d, act_inv_scale [B]) with the sample's power-of-two range normalisation folded
    into a and d (split-operand convs: conv3d_gcr_split undoes it in the epilogue)"""
    s0, q0, V0 = st0
    B, C0 = s0.shape
    if st1 is not None:
        s1, q1, V1 = st1
        C1 = s1.shape[1]
        rep = V0 // V1
    else:
        s1 = q1 = None
        C1, V1, rep = 0, 0, 1
    a = torch.empty((B, C0 + C1), dtype=torch.float32, device=s0.device)
    d = torch.empty_like(a)
    inv = torch.empty(B, dtype=torch.float32, device=s0.device) if with_act_scale else None
    _lib.call("gn_groupnorm_affine", _p(s0), _p(q0), C0, V0, _p(s1), _p(q1), C1, V1, rep, B, groups, float(eps), _p(gamma), _p(beta),
              _p(a), _p(d), _p(inv), _stream())
    return (a, d, inv) if with_act_scale else (a, d)


def pack_conv_weight(w):
    """nn.Conv3d weight (Cout, Cin, 3, 3, 3) -> [27 taps][Cin/16][Cout][16] (the B-operand pack of gn_conv3d_gcr)."""
    cout, cin = w.shape[:2]
    assert cin % 16 == 0
    return w.detach().float().permute(2, 3, 4, 1, 0).reshape(27, cin // 16, 16, cout).permute(0, 1, 3, 2).contiguous()


from .arith import CONV_FP32, CONV_MODE_NAMES, SPLIT_BF16X2, SPLIT_BF16X3, SPLIT_F16X2  # noqa: E402,F401  (which arithmetic runs is an arith.Arith carried by the call)


class SplitPack:
    """weight planes of the split-precision conv: .tensor (int16 bit patterns, MFMA-fragment order), .mode, .out_scale (fp32 [Cout]:
    the exact powers of two that undo the per-output-channel weight scales)"""

    def __init__(self, tensor, mode, out_scale):
        self.tensor, self.mode, self.out_scale = tensor, int(mode), out_scale

    def to(self, device):
        return SplitPack(self.tensor.to(device), self.mode, self.out_scale.to(device))


def pack_conv_weight_split(w, mode):
    """(Cout, Cin, 3,3,3) fp32 -> exact plane decomposition (w = w1 + w2 [+ w3], residual chain) in MFMA-fragment order
    [Cin/16][27][Cout/32][planes][h 2][r 32][8] (lane 32h+r holds channels 8h..8h+7 of cout 32*blk+r), followed by eight zero
    (slice, tap) steps: the kernels' fragment DMA runs up to two groups of three steps ahead of the last one.  mode SPLIT_BF16X2/3: bf16 planes;
    SPLIT_F16X2: two fp16 planes of w * 2^k(cout), k chosen PER OUTPUT CHANNEL so that the row maximum max|w[cout]| * 2^k is in [1, 2)
    (out_scale[cout] = 2^-k undoes it exactly): every weight within 2^3 of its row's largest keeps a normal second plane (residual
    <= 2^-22 |w|), smaller ones are carried to 2^-25 of the row maximum -- always far below the row's own dot-product magnitude,
    also for a heavy-tailed trained tensor (a per-TENSOR scale would push whole rows into the subnormal second plane)."""
    if mode not in (SPLIT_BF16X2, SPLIT_BF16X3, SPLIT_F16X2):
        raise ValueError(f"unknown split mode {mode}")
    cout, cin = w.shape[:2]
    w = w.detach().float()
    planes, dt = (2, torch.float16) if mode == SPLIT_F16X2 else (int(mode), torch.bfloat16)
    scale = torch.ones(cout, dtype=torch.float32)
    if mode == SPLIT_F16X2:
        m = w.reshape(cout, -1).abs().amax(dim=1).cpu()
        ok = torch.isfinite(m) & (m > 0)
        scale = torch.where(ok, torch.exp2(-torch.floor(torch.log2(torch.where(ok, m, torch.ones_like(m))))), scale).float()
    base = (w * scale.to(w.device).view(cout, 1, 1, 1, 1)).permute(2, 3, 4, 1, 0).reshape(27, cin // 16, 2, 8, cout // 32, 32)                # [tap][S][h][8][blk][r]
    base = base.permute(1, 0, 4, 2, 5, 3)                                                                # [S][tap][blk][h][r][8]
    out, r = [], base
    for _ in range(planes):
        p = r.to(dt)
        out.append(p)
        r = r - p.float()
    pk = torch.stack(out, dim=3).contiguous()                                                            # [S][tap][blk][planes][h][r][8]
    pk = pk.reshape(cin // 16 * 27, -1)
    pk = torch.cat([pk, torch.zeros_like(pk[:1]).repeat(8, 1)], dim=0)
    return SplitPack(pk.contiguous().view(torch.int16), mode, (1.0 / scale).contiguous().to(w.device))


def wino_supported(cin, cout, dims):
    """shapes gn_conv3d_gcr_split_wino takes: one source, 32-bit byte offsets inside a sample, and whole 4 x 8 x 8 tiles with Cin <= 256 for the 128-wide
    kernel (Cout % 128 == 0, csrc/unet_wino.hip) / whole 8 x 8 x 8 tiles with Cin <= 128 for the 32-wide column-block kernel (csrc/unet_wino32.hip)"""
    D, H, W = [int(v) for v in dims]
    if not (cin % 16 == 0 and cout % 32 == 0 and H % 8 == 0 and W % 8 == 0 and D * H * W <= (1 << 27) and D * H * W * cin * 4 < (1 << 32)):
        return False
    return (cin <= 256 and D % 4 == 0) if cout % 128 == 0 else (cin <= 128 and D % 8 == 0)


def pack_conv_weight_split_wino(w):
    """(Cout, Cin, 3,3,3) fp32 -> the Winograd F(2,3)-along-x pack of gn_conv3d_gcr_split_wino (csrc/unet_wino.hip): per (kd, kh, channel) the three
    kw taps g0 g1 g2 become the four transform positions (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2), in fp64; per-output-channel power-of-two
    scale over the TRANSFORMED row (row maximum in [1, 2)), two fp16 planes, fragment order [Cin/16][36 steps = (j * 3 + kd) * 3 + kh][Cout/32][plane]
    [h 2][r 32][8] + six zero steps (the kernel's fragment DMA runs two groups of three steps ahead)."""
    cout, cin = w.shape[:2]
    w = w.detach().double().cpu()
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    u = torch.einsum("jk,nczyk->nczyj", G, w)                                                           # (Cout, Cin, kd, kh, j)
    m = u.reshape(cout, -1).abs().amax(dim=1)
    scale = _pow2_floor_inv(m)
    base = (u * scale.view(cout, 1, 1, 1, 1)).float().permute(4, 2, 3, 1, 0).reshape(36, cin // 16, 2, 8, cout // 32, 32)   # [step][S][h][8][blk][r]
    base = base.permute(1, 0, 4, 2, 5, 3)                                                               # [S][step][blk][h][r][8]
    p1 = base.to(torch.float16)
    p2 = (base - p1.float()).to(torch.float16)
    pk = torch.stack((p1, p2), dim=3).contiguous().reshape(cin // 16 * 36, -1)                          # [S * step][blk][plane][h][r][8]
    pk = torch.cat([pk, torch.zeros_like(pk[:6])], dim=0)
    return SplitPack(pk.contiguous().view(torch.int16), SPLIT_F16X2, (1.0 / scale).float().contiguous())


def conv3d_gcr_split_wino(src, a, d, pack, cout, relu=True, with_stats=False, act_inv=None, tile_active=None, kconst=None, kreach=1, partial=None):
    """the literal form through the Winograd kernels (pack = pack_conv_weight_split_wino); partial: the polyphase partial of a decoder's first
    convolution (upconv_partial), 32- / 64-wide layers only"""
    B, D, H, W, C0 = src.shape
    out = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=src.device)
    s, q = _stats_buffers(B, cout, src.device, with_stats)
    if partial is not None:
        if tile_active is not None:
            raise ValueError("conv3d_gcr_split_wino: the occupancy-aware launch cannot take a polyphase partial")
        _lib.call("gn_conv3d_gcr_split_wino_partial", _p(src), C0, _p(a), _p(d), _p(pack.tensor), _p(pack.out_scale), _p(act_inv), None, B, D, H, W, cout,
                  1 if relu else 0, _p(out), _p(s), _p(q), _p(_chk(partial, torch.float32, "partial")), _stream())
        return (out, (s, q, D * H * W)) if with_stats else out
    ows, ows_bytes = _occupancy_ws(tile_active, B, D, H, W, src.device)
    _lib.call("gn_conv3d_gcr_split_wino", _p(src), C0, _p(a), _p(d), _p(pack.tensor), _p(pack.out_scale), _p(act_inv), None, B, D, H, W, cout,
              1 if relu else 0, _p(out), _p(s), _p(q), _p(tile_active), _p(kconst), int(kreach), _p(ows), ows_bytes, _stream())
    return (out, (s, q, D * H * W)) if with_stats else out


def grid_tile_flags(flat_idx, B, grid_shape, reach=1):
    """uint8 [B][tiles]: the 4 x 8 x 8 output tiles of a 3x3x3 conv over the scattered volume that can see an occupied cell (reach 1),
    or of the conv behind it (reach 2)"""
    g0, g1, g2 = [int(v) for v in grid_shape]
    tiles = -(-g0 // 4) * -(-g1 // 8) * -(-g2 // 8)
    flags = torch.empty((B, tiles), dtype=torch.uint8, device=flat_idx.device)
    _lib.call("gn_grid_tile_flags", _p(_chk(flat_idx, _i32, "flat_idx")), flat_idx.numel(), B, g0, g1, g2, int(reach), _p(flags), _stream())
    return flags


def polyphase_weights(w, c0):
    """nn.Conv3d weight (Cout, C0 + C1, 3,3,3) of a layer that reads cat((skip [C0 ch], upsample_nearest_x2(x) [C1 ch])) ->
    (w0 (Cout, C0, 3,3,3): the full-resolution part unchanged;
     wm (8 * Cout, C1, 3,3,3): the upsampled part as a convolution over the COARSE volume, output channel (class, n), class = 4 pz + 2 py + px
         the parity of the fine output voxel: a fine tap d in {-1, 0, +1} of an even voxel lands on coarse offset {-1, 0, 0}, of an odd voxel
         on {0, 0, +1}, so the 27 fine taps merge (sums of weights, fp64) into 2 x 2 x 2 coarse taps per class -- exact algebra;
     tapmask int32 [27]: bit i = 32-wide output block i of wm has a non-zero weight at tap (kd * 3 + kh) * 3 + kw -- diagnostic: every
         class uses exactly 2 x 2 x 2 of the 27 coarse taps)."""
    w = w.detach().double().cpu()
    cout = w.shape[0]
    w0, w1 = w[:, :c0].float(), w[:, c0:]
    M = torch.zeros(2, 3, 3, dtype=torch.float64)            # [parity][coarse tap][fine tap]
    M[0, 0, 0] = M[0, 1, 1] = M[0, 1, 2] = 1.0
    M[1, 1, 0] = M[1, 1, 1] = M[1, 2, 2] = 1.0
    blocks = [torch.einsum("az,by,cx,nkzyx->nkabc", M[pz], M[py], M[px], w1) for pz in (0, 1) for py in (0, 1) for px in (0, 1)]
    wm = torch.cat(blocks, dim=0).float()                    # (8 * Cout, C1, 3, 3, 3)
    nz = (wm.reshape(8 * cout // 32, 32, -1, 27) != 0).any(dim=2).any(dim=1)       # [block][tap]
    mask = torch.zeros(27, dtype=torch.int64)
    for blk in range(nz.shape[0]):
        mask |= nz[blk].to(torch.int64) << blk
    return w0.contiguous(), wm.contiguous(), mask.to(torch.int32)


def pack_upconv_weight(wm, cout, mode):
    """merged polyphase weights wm (8 * cout, C1, 3,3,3) of polyphase_weights -> SplitPack for gn_upconv_partial: the 8 taps class
    (pz, py, px) uses are wm[..., pz + iz, py + iy, px + ix], i in {0,1}; fragment order [C1/16][tap = 4 iz + 2 iy + ix][class][cout/32]
    [plane][h][r][8] (lane 32 h + r holds channels 8h..8h+7 of output r of the block), per-row power-of-two scale for fp16 planes."""
    if mode not in (SPLIT_BF16X2, SPLIT_F16X2):
        raise ValueError("gn_upconv_partial runs the two-plane modes only")
    wm = wm.detach().float().cpu()
    c1 = wm.shape[1]
    assert wm.shape[0] == 8 * cout and c1 % 16 == 0 and cout % 32 == 0
    taps = torch.empty((8, 8, cout, c1), dtype=torch.float32)                                     # [class][tap][n][k]
    for c in range(8):
        pz, py, px = c >> 2, (c >> 1) & 1, c & 1
        for t in range(8):
            iz, iy, ix = t >> 2, (t >> 1) & 1, t & 1
            taps[c, t] = wm[c * cout:(c + 1) * cout, :, pz + iz, py + iy, px + ix]
    scale = torch.ones(8, cout)
    if mode == SPLIT_F16X2:
        m = taps.abs().amax(dim=(1, 3))
        ok = torch.isfinite(m) & (m > 0)
        scale = torch.where(ok, torch.exp2(-torch.floor(torch.log2(torch.where(ok, m, torch.ones_like(m))))), scale)
    dt = torch.float16 if mode == SPLIT_F16X2 else torch.bfloat16
    t_ = (taps * scale[:, None, :, None]).reshape(8, 8, cout // 32, 32, c1 // 16, 2, 8)            # [class][tap][blk][r][S][h][i]
    t_ = t_.permute(4, 1, 0, 2, 5, 3, 6)                                                          # [S][tap][class][blk][h][r][i]
    p1 = t_.to(dt)
    p2 = (t_ - p1.float()).to(dt)
    pk = torch.stack((p1, p2), dim=4).contiguous()                                                # [S][tap][class][blk][plane][h][r][i]
    return SplitPack(pk.view(torch.int16).reshape(-1), mode, (1.0 / scale).reshape(-1).contiguous())


def upconv_partial(src1, a1, d1, pack, cout, act_inv=None):
    """coarse source [B][Dc][Hc][Wc][C1] -> polyphase partial sums [B][Dc][Hc][Wc][8 * cout] (csrc/upconv.hip)"""
    B, Dc, Hc, Wc, C1 = src1.shape
    part = torch.empty((B, Dc, Hc, Wc, 8 * cout), dtype=torch.float32, device=src1.device)
    _lib.call("gn_upconv_partial", _p(src1), C1, _p(_chk(a1, torch.float32, "a")), _p(_chk(d1, torch.float32, "d")), _p(pack.tensor), pack.mode,
              _p(pack.out_scale), _p(act_inv), B, Dc, Hc, Wc, cout, _p(part), _stream())
    return part


def conv3d_gcr_split(src0, src1, a, d, pack, cout, relu=True, with_stats=False, act_inv=None, tile_active=None, kconst=None, kreach=1,
                     partial=None):
    B, D, H, W, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[-1]
    out = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=src0.device)
    s, q = _stats_buffers(B, cout, src0.device, with_stats)
    ows, ows_bytes = _occupancy_ws(tile_active, B, D, H, W, src0.device)
    _lib.call("gn_conv3d_gcr_split", _p(src0), C0, _p(src1), C1, _p(a), _p(d), _p(pack.tensor), pack.mode, _p(pack.out_scale), _p(act_inv), B, D, H, W, cout,
              1 if relu else 0, _p(out), _p(s), _p(q), _p(tile_active), _p(kconst), int(kreach), _p(partial), _p(ows), ows_bytes, _stream())
    return (out, (s, q, D * H * W)) if with_stats else out


class AffinePack:
    """what gn_conv_affine_pack prepared for ONE batch of one 'gcr' layer: per-sample fp16x2 weight sets with the GroupNorm affine folded
    in, the staging affine (s, -c s), the output scales and the border-class bias table"""
    __slots__ = ("pack", "stage_a", "stage_d", "out_scale", "kbias", "cin", "cout", "wino")


def conv_affine_pack(weight, a, d, stats, rest=None, wino=False):
    """weight: the raw Conv3d weight [Cout][Cin][3][3][3] (fp32, device); a, d [B][Cin]: the GroupNorm affine (groupnorm_affine without the
    sample scale); stats = (sum, sumsq, V) of the layer's input; rest [B][Cin] or None (zeros): the value the input holds away from its
    support -> AffinePack (csrc/conv_prep.hip).  wino: the pack holds the Winograd F(2,3)-along-x transformed weights (csrc/unet_wino.hip)"""
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    B, dev = a.shape[0], a.device
    w = weight.detach()
    _chk(w, torch.float32, "weight")
    _chk(a, torch.float32, "a"); _chk(d, torch.float32, "d")
    if rest is not None:
        _chk(rest, torch.float32, "rest")
        assert tuple(rest.shape) == (B, cin)
    sfx = "_wino" if wino else ""
    nbytes = getattr(_lib.load(), "gn_conv_affine_pack" + sfx + "_bytes")(B, cin, cout)
    if nbytes == 0 and B > 0:
        raise ValueError("conv_affine_pack: channel counts must be multiples of 16 (in) / 32 (out)")
    r = AffinePack()
    r.cin, r.cout, r.wino = cin, cout, bool(wino)
    r.pack = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
    r.stage_a = torch.empty((B, cin), dtype=torch.float32, device=dev)
    r.stage_d = torch.empty((B, cin), dtype=torch.float32, device=dev)
    r.out_scale = torch.empty((B, cout), dtype=torch.float32, device=dev)
    r.kbias = torch.empty((B, 64, cout), dtype=torch.float32, device=dev)
    ws = torch.empty((B * cin * 12 + B * cout * 4 + 16,), dtype=torch.uint8, device=dev)
    _lib.call("gn_conv_affine_pack" + sfx, _p(w), cin, cout, _p(a), _p(d), _p(stats[0]), _p(stats[1]), int(stats[2]), _p(rest), B, _p(r.pack), nbytes,
              _p(r.stage_a), _p(r.stage_d), _p(r.out_scale), _p(r.kbias), _p(ws), ws.numel(), _stream())
    return r


def conv3d_gcr_split_persample(src, prep, relu=True, with_stats=False, tile_active=None, kconst=None, kreach=1, partial=None):
    """the 'gcr' layer from an AffinePack (GN_SPLIT_F16X2 arithmetic; the operand is exactly zero wherever the input is at rest)"""
    B, D, H, W, C = src.shape
    assert C == prep.cin and B == prep.stage_a.shape[0]
    out = torch.empty((B, D, H, W, prep.cout), dtype=torch.float32, device=src.device)
    s, q = _stats_buffers(B, prep.cout, src.device, with_stats)
    ows, ows_bytes = _occupancy_ws(tile_active, B, D, H, W, src.device)
    if prep.wino:
        if partial is not None:
            if tile_active is not None:
                raise ValueError("conv3d_gcr_split_persample: the occupancy-aware launch cannot take a polyphase partial")
            _lib.call("gn_conv3d_gcr_split_wino_partial", _p(src), C, _p(prep.stage_a), _p(prep.stage_d), _p(prep.pack), _p(prep.out_scale), None, _p(prep.kbias),
                      B, D, H, W, prep.cout, 1 if relu else 0, _p(out), _p(s), _p(q), _p(_chk(partial, torch.float32, "partial")), _stream())
            return (out, (s, q, D * H * W)) if with_stats else out
        _lib.call("gn_conv3d_gcr_split_wino", _p(src), C, _p(prep.stage_a), _p(prep.stage_d), _p(prep.pack), _p(prep.out_scale), None, _p(prep.kbias),
                  B, D, H, W, prep.cout, 1 if relu else 0, _p(out), _p(s), _p(q), _p(tile_active), _p(kconst), int(kreach), _p(ows), ows_bytes, _stream())
        return (out, (s, q, D * H * W)) if with_stats else out
    _lib.call("gn_conv3d_gcr_split_persample", _p(src), C, _p(prep.stage_a), _p(prep.stage_d), _p(prep.pack), _p(prep.out_scale), _p(prep.kbias),
              B, D, H, W, prep.cout, 1 if relu else 0, _p(out), _p(s), _p(q), _p(tile_active), _p(kconst), int(kreach), _p(partial), _p(ows), ows_bytes,
              _stream())
    return (out, (s, q, D * H * W)) if with_stats else out


def _occupancy_ws(tile_active, B, D, H, W, device):
    """workspace of an occupancy-aware conv launch (the compact list of active tiles): (tensor | None, bytes)"""
    if tile_active is None:
        return None, 0
    n = _lib.load().gn_conv3d_occupancy_workspace_bytes(B, D, H, W)
    return torch.empty((n,), dtype=torch.uint8, device=device), n


def _stats_buffers(B, C, device, want):
    if not want:
        return None, None
    sq = torch.empty((2, B, C), dtype=torch.float64, device=device)      # back to back: the library zeroes the pair with one fill (gn_zero_stats)
    return sq[0], sq[1]


def conv3d_gcr(src0, src1, a, d, wp, cout, relu=True, with_stats=False):
    """-> out, or (out, (sum, sumsq, V)) with the statistics of the output when with_stats"""
    B, D, H, W, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[-1]
    out = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=src0.device)
    s, q = _stats_buffers(B, cout, src0.device, with_stats)
    _lib.call("gn_conv3d_gcr", _p(src0), C0, _p(src1), C1, _p(a), _p(d), _p(wp), B, D, H, W, cout, 1 if relu else 0, _p(out),
              _p(s), _p(q), _stream())
    return (out, (s, q, D * H * W)) if with_stats else out


ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3


def affine_act(x, a=None, d=None, bias=None, act=ACT_NONE, out=None):
    """y = act(x * a[b][c] + d[b][c] + bias[c]) over a channel-last volume [B][...][C] (gn_affine_act: the non-'gcr' layer orders' odds and ends);
    in place when out is x"""
    B, C = x.shape[0], x.shape[-1]
    V = x.numel() // max(B * C, 1)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("gn_affine_act", _p(x), B, V, C, _p(a), _p(d), _p(bias), int(act), _p(out), _stream())
    return out


def maxpool3d_2(x, with_stats=False):
    B, D, H, W, C = x.shape
    out = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    s, q = _stats_buffers(B, C, x.device, with_stats)
    _lib.call("gn_maxpool3d_2", _p(x), B, D, H, W, C, _p(out), _p(s), _p(q), _stream())
    return (out, (s, q, (D // 2) * (H // 2) * (W // 2))) if with_stats else out


# ------------------------------------------------------------------------------------------------ decoder
def trilinear_sample(vol_b, query=None, Q=0, m0=0, M=None, out=None):
    """vol_b: one sample, channel-last [D][H][W][C].  query [M][3] or lattice rows m0..m0+M of (Q,Q,Q)."""
    D, H, W, C = vol_b.shape
    if query is not None:
        M = query.shape[0]
        _chk(query, torch.float32, "query")
    if out is None:
        out = new_rows(M, C, vol_b.device)
    _lib.call("gn_trilinear_sample", _p(vol_b), D, H, W, C, _p(query), int(Q), int(m0), int(M), _p(out), rows_view(out)[1], _stream())
    return out


def pack_kpair(w):
    """Linear weight W[n][k] -> Wp[k/16][n][2][8] with Wp[g][n][h][j] = W[n][16g + 8h + j]: the B-operand pack of
    gn_implicit_decode (32 contiguous bytes per lane, 2 KB per wave per 16-deep k-group)."""
    n, k = w.shape
    assert k % 16 == 0
    return w.detach().float().reshape(n, k // 16, 2, 8).permute(1, 0, 2, 3).contiguous()


def implicit_decode(vol_b, layers, query=None, Q=0, m0=0, M=None, out=None, xin=None, run_if=None):
    """vol_b [D][H][W][C0]; layers = ((w1p,b1,s1,t1,N1), (w2p,b2,s2,t2,N2), (w3,b3,s3,t3,OUT)) -> out [M][OUT].
    xin: optional pre-sampled rows [M][C0] (then only the MLP runs).  run_if: optional device float; the launch is a no-op unless it
    is non-zero (the fp32 twin of a gated implicit_decode_split call)."""
    (w1p, b1, s1, t1, N1), (w2p, b2, s2, t2, N2), (w3, b3, s3, t3, OUT) = layers
    if xin is not None:
        D = H = W = 0
        M, C0 = xin.shape
        ldxin = rows_view(xin)[1]
    else:
        D, H, W, C0 = vol_b.shape
        ldxin = 0
    if query is not None:
        M = query.shape[0]
        _chk(query, torch.float32, "query")
    if out is None:
        out = torch.empty((M, OUT), dtype=torch.float32, device=(xin if xin is not None else vol_b).device)
    _lib.call("gn_implicit_decode", _p(vol_b), D, H, W, C0, _p(xin), ldxin, _p(query), int(Q), int(m0), int(M), _p(w1p), _p(b1), _p(s1), _p(t1), N1,
              _p(w2p), _p(b2), _p(s2), _p(t2), N2, _p(w3), _p(b3), _p(s3), _p(t3), OUT, _p(out), rows_view(out)[1], _p(run_if), _stream())
    return out


def trilinear_sample_batch(vol, query, out=None):
    """vol (B, D, H, W, C) channel-last, query (B, M, 3) -> rows (B, M, C) [leading dimension padded to 4]: every volume's queries in one launch"""
    B, D, H, W, C = vol.shape
    M = query.shape[1]
    _chk(vol, torch.float32, "vol")
    _chk(query, torch.float32, "query")
    if out is None:
        out = torch.empty((B, M, pad4(C)), dtype=torch.float32, device=vol.device)[:, :, :C]
    _lib.call("gn_trilinear_sample_batch", _p(vol), B, vol.stride(0), D, H, W, C, _p(query), int(M), _p(out), out.stride(1), _stream())
    return out


def implicit_decode_batch(xin, layers, out, run_if=None, run_if_stride=0):
    """implicit_decode(xin=...) for B row sets in one launch: xin (B, M, C0) [row stride = leading dimension], out (B, M, OUT); run_if: one device flag per
    row set at run_if[b * run_if_stride] (the gated fp32 twin of implicit_decode_split_batch)"""
    (w1p, b1, s1, t1, N1), (w2p, b2, s2, t2, N2), (w3, b3, s3, t3, OUT) = layers
    B, M, C0 = xin.shape
    _lib.call("gn_implicit_decode_batch", _p(xin), xin.stride(1), int(M), B, C0, _p(w1p), _p(b1), _p(s1), _p(t1), N1, _p(w2p), _p(b2), _p(s2), _p(t2), N2,
              _p(w3), _p(b3), _p(s3), _p(t3), OUT, _p(out), out.stride(1), _p(run_if), int(run_if_stride), _stream())
    return out


def implicit_decode_split_batch(xin, pack, out, xscale=None):
    """implicit_decode_split for B row sets in ONE launch: xin (B, M, C0), out (B, M, OUT), xscale None or (B, 4) -- row set b with its own input scale
    record; row for row the results of the single call (the persistent workgroups are shared between the sets through blockIdx.y)"""
    B, M, C0 = xin.shape
    assert xin.stride(0) == M * xin.stride(1) and out.stride(0) == M * out.stride(1), "row sets must be packed back to back"
    _lib.call("gn_implicit_decode_split_batch", _p(xin), xin.stride(1), int(M), B, _p(pack.wpack), _p(pack.tab), _p(xscale),
              C0, pack.hidden, pack.hidden, pack.out_channels, _p(out), out.stride(1), _stream())
    return out


class DecodeSplitPack:
    """weights / epilogue tables of gn_implicit_decode_split for one [128 | 32, 256, 256, OUT] decoder; smax: upper bound for the run-time
    input scale (keeps the scaled biases small, see pack_decode_split)"""

    def __init__(self, wpack, tab, smax, out_channels, hidden=256):
        self.wpack, self.tab, self.smax, self.out_channels, self.hidden = wpack, tab, float(smax), int(out_channels), int(hidden)

    def to(self, device):
        return DecodeSplitPack(self.wpack.to(device), self.tab.to(device), self.smax, self.out_channels, self.hidden)


def _pow2_floor_inv(m):
    """per-row power of two r with r * m in [1, 2) (1 where m is 0 / non-finite); m: fp64 tensor"""
    ok = torch.isfinite(m) & (m > 0)
    return torch.where(ok, torch.exp2(-torch.floor(torch.log2(torch.where(ok, m, torch.ones_like(m))))), torch.ones_like(m))


def _d_unit(q, h):
    """hidden unit (within a 32-unit block) held by accumulator register q of lane half h (32x32 MFMA D layout)"""
    return (q & 3) + 8 * (q >> 2) + 4 * h


def pack_decode_split(layers):
    """layers = ((w1,b1,s1,t1), (w2,b2,s2,t2), (w3,b3,s3,t3)) with w1 (N,128) or (N,32), w2 (N,N), w3 (OUT,N) fp32, N = 256 (the shipped decoders) or 512
    (the class default, networks/conv_implicit_wnf.py:122; 32-channel folded input only), s/t = folded BatchNorm scale/shift or None -> DecodeSplitPack.
    The BatchNorm affine of hidden layer i is folded into layer i+1 (W' = W diag(s), b' = b + W t, in fp64).  Every hidden unit j carries a static
    power-of-two scale r_j (its weight row's maximum scaled into [1, 2)): the kernel keeps hidden activations in those units (r_j * s_x * h_j, s_x =
    the garment's run-time input scale) and 1 / r_j is folded into the next layer's column j -- all exact.  Weight stages of 16 KB:
    [4 k-groups][2 blocks][2 planes][64 lanes][8 fp16], steps in (block pair, k-group) order, layer 1 then layer 2; layer 2's k order follows the
    register layout the layer-1 accumulators already have (see csrc/decode_split.hip).  w1 (N, 32): the first layer with the UNet's final 1x1x1
    convolution folded in (ImplicitWNFDecoder.folded_pack)."""
    (w1, b1, s1, t1), (w2, b2, s2, t2), (w3, b3, s3, t3) = layers
    dd = lambda v, n, fill: (torch.full((n,), fill, dtype=torch.float64) if v is None else v.detach().double().cpu())
    w1, w2, w3 = w1.detach().double().cpu(), w2.detach().double().cpu(), w3.detach().double().cpu()
    out_c, N = w3.shape[0], w2.shape[0]
    assert N in (256, 512) and w1.shape[0] == N and w2.shape == (N, N) and w3.shape[1] == N and 1 <= out_c <= 4
    assert w1.shape[1] in ((128, 32) if N == 256 else (32,)), "the 512-wide pack takes the folded 32-channel first layer"
    k0g, npair, nb_, kg2 = w1.shape[1] // 16, N // 64, N // 32, N // 16                           # 16-deep k-groups of layer 1; block pairs, blocks, k-groups of layer 2
    b2f = dd(b2, N, 0.0) + w2 @ dd(t1, N, 0.0)
    w2 = w2 * dd(s1, N, 1.0)[None, :]
    b3f = (dd(b3, out_c, 0.0) + w3 @ dd(t2, N, 0.0)).float()
    w3 = w3 * dd(s2, N, 1.0)[None, :]
    b1f = dd(b1, N, 0.0)
    # per-unit scales (exact powers of two); a weight or bias that leaves fp32's range through them would be a broken checkpoint anyway
    r1 = _pow2_floor_inv(w1.abs().amax(dim=1))
    w1s, b1s = (w1 * r1[:, None]).float(), (b1f * r1).float()
    w2 = w2 / r1[None, :]
    r2 = _pow2_floor_inv(w2.abs().amax(dim=1))
    w2s, b2s = (w2 * r2[:, None]).float(), (b2f * r2).float()
    w3s = (w3 / r2[None, :]).float()
    # run-time input scale s_x <= smax keeps every scaled bias below 2^13 (hidden values = accumulator + bias stay inside fp16)
    bmax = max(float(b1s.abs().max()), float(b2s.abs().max()))
    smax = 2.0 ** 60 if not (bmax > 0 and math.isfinite(bmax)) else min(2.0 ** 60, 2.0 ** math.floor(math.log2(8192.0 / bmax)))
    ar = torch.arange
    bp, g1, blk, h, r, i = torch.meshgrid(ar(npair), ar(k0g), ar(2), ar(2), ar(32), ar(8), indexing="ij")
    a1 = w1s[32 * (2 * bp + blk) + r, 16 * g1 + 8 * h + i]                                       # [pair][k-group][blk][h][r][i]
    bp, g2, blk, h, r, i = torch.meshgrid(ar(npair), ar(kg2), ar(2), ar(2), ar(32), ar(8), indexing="ij")
    q = 8 * (g2 & 1) + i
    a2 = w2s[32 * (2 * bp + blk) + r, 32 * (g2 >> 1) + (q & 3) + 8 * (q >> 2) + 4 * h]          # [pair][k-group][blk][h][r][i]

    def planes(a):                                   # [..steps..][blk][h][r][i] -> [stage][4 steps][blk][plane][h][r][i]
        p1 = a.to(torch.float16)
        p2 = (a - p1.float()).to(torch.float16)
        st = torch.stack((p1, p2), dim=-4)                                                        # [...][blk][plane][h][r][i]
        return st.reshape(-1, 4, 2, 2, 2, 32, 8)                                                  # steps in (pair, k-group) order, 4 per stage

    wpack = torch.cat((planes(a1), planes(a2)), dim=0).contiguous().view(torch.int16)             # [(npair * (k0g + kg2)) / 4 stages][...]
    assert wpack.numel() * 2 == npair * (k0g + kg2) // 4 * 16384
    nb, hh, qq = torch.meshgrid(ar(nb_), ar(2), ar(16), indexing="ij")
    u = 32 * nb + (qq & 3) + 8 * (qq >> 2) + 4 * hh                                               # [blocks][2][16]
    tab1 = b1s[u]
    tab2 = torch.stack([b2s[u]] + [w3s[o][u] for o in range(out_c)], dim=2)                       # [blocks][2][1+OUT][16]
    tail = torch.stack((b3f, dd(s3, out_c, 1.0).float(), dd(t3, out_c, 0.0).float()))
    tab = torch.cat((tab1.reshape(-1), tab2.reshape(-1), tail.reshape(-1))).float().contiguous()
    return DecodeSplitPack(wpack, tab, smax, out_c, hidden=N)


def decoder_input_scale(sumsq, V, smax):
    """per-garment input scale of gn_implicit_decode_split from the per-(sample, channel) sums of squares of the sampled volume
    -> float32 [B][4] = (s, 1/s, unsafe, 0); unsafe = 1: the garment goes to the gated fp32 kernel (decided on the device)"""
    B, C = sumsq.shape
    out = torch.empty((B, 4), dtype=torch.float32, device=sumsq.device)
    _lib.call("gn_decoder_input_scale", _p(_chk(sumsq, torch.float64, "sumsq")), int(V), B, C, float(smax), _p(out), _stream())
    return out


def implicit_decode_split(xin, pack, out=None, xscale=None):
    """pre-sampled rows xin [M][128 | 32] -> out [M][OUT] through the [128 | 32, 256, 256, OUT] decoder on the 16-bit matrix cores.
    xscale: the garment's (s, 1/s, unsafe, 0) row of decoder_input_scale (None: unscaled rows -- fine for O(1) inputs only); with
    unsafe != 0 the launch is a no-op and the caller's implicit_decode(..., run_if=xscale[2:3]) fills `out`"""
    M, C0 = xin.shape
    if out is None:
        out = torch.empty((M, pack.out_channels), dtype=torch.float32, device=xin.device)
    _lib.call("gn_implicit_decode_split", _p(xin), rows_view(xin)[1], int(M), _p(pack.wpack), _p(pack.tab), _p(xscale),
              C0, pack.hidden, pack.hidden, pack.out_channels, _p(out), rows_view(out)[1], _stream())
    return out


def implicit_decode_lattice_split(vol_b, Q, pack, out, xscale=None, m0=0, M=None):
    """rows m0 .. m0+M-1 of the (Q,Q,Q) lattice through the folded scalar decoder, sampled INSIDE the decoder kernel from the channel-last
    volume vol_b [D][H][W][32] (no sampled-row buffer); bit-identical to trilinear_sample + implicit_decode_split"""
    D, H, W, C0 = vol_b.shape
    M = Q * Q * Q - m0 if M is None else M
    _lib.call("gn_implicit_decode_lattice_split", _p(_chk(vol_b, torch.float32, "vol")), D, H, W, C0, int(Q), int(m0), int(M), _p(pack.wpack), _p(pack.tab),
              _p(xscale), 256, 256, pack.out_channels, _p(out), rows_view(out)[1], _stream())
    return out


def lattice_split_supported(vol_b, pack):
    D, H, W, C0 = vol_b.shape
    return C0 == 32 and pack.out_channels == 1 and pack.hidden == 256 and D * H * W * 128 < 2 ** 32 and vol_b.is_contiguous()


# ------------------------------------------------------------------------------------------------ isosurface
def _ggm_tmp(shape, sigma, device):
    """the 8-pass form's workspace; None for the fused launch (kernel radius <= 2: csrc/iso.hip GGM_R)"""
    return None if int(4.0 * float(sigma) + 0.5) <= 2 else torch.empty((2,) + tuple(shape), dtype=torch.float32, device=device)


def ggm3d(vol, sigma):
    _chk(vol, torch.float32, "vol")
    n0, n1, n2 = vol.shape
    tmp = _ggm_tmp(vol.shape, sigma, vol.device)
    out = torch.empty_like(vol)
    _lib.call("gn_ggm3d", _p(vol), n0, n1, n2, float(sigma), _p(tmp), _p(out), _stream())
    return out


def minmax(x):
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.call("gn_minmax", _p(_chk(x, torch.float32, "x")), x.numel(), _p(out), _stream())
    return out


def ggm3d_batch(vols, sigma):
    """ggm3d of every (n0,n1,n2) volume of a (B,n0,n1,n2) batch in one set of launches"""
    _chk(vols, torch.float32, "vols")
    B, n0, n1, n2 = vols.shape
    tmp = _ggm_tmp(vols.shape, sigma, vols.device)
    out = torch.empty_like(vols)
    _lib.call("gn_ggm3d_batch", _p(vols), B, n0, n1, n2, float(sigma), _p(tmp), _p(out), _stream())
    return out


def ggm3d_batch_range(vols, sigma, accum_bits=64):
    """ggm3d_batch + the (B,2) float32 (min, max) record of every volume from the same launch (NaN-propagating like numpy's; no pass of its own
    over the volumes: gn_ggm3d_batch_ex).  accum_bits=32: the taps accumulate in fp32 (no scipy bit-parity; Arith.ggm_fp32)"""
    _chk(vols, torch.float32, "vols")
    B, n0, n1, n2 = vols.shape
    tmp = _ggm_tmp(vols.shape, sigma, vols.device)
    out = torch.empty_like(vols)
    rng = torch.empty((B, 2), dtype=torch.float32, device=vols.device)
    nws = _lib.load().gn_ggm3d_range_workspace_bytes(B, n0, n1, n2)             # a (min, max) pair per wave of the fused launch; scratch
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=vols.device)
    _lib.call("gn_ggm3d_batch_ex", _p(vols), B, n0, n1, n2, float(sigma), _p(tmp), _p(out), int(accum_bits), _p(rng), _p(ws), nws, _stream())
    return out, rng


def minmax_batch(vols):
    """-> (B,2) float32: (min, max) of every volume of a (B,...) batch (NaN-propagating: a NaN anywhere gives NaN, as numpy.min / numpy.max)"""
    _chk(vols, torch.float32, "vols")
    B = vols.shape[0]
    out = torch.empty((B, 2), dtype=torch.float32, device=vols.device)
    if B:
        _lib.call("gn_minmax_batch", _p(vols), B, vols.numel() // B, _p(out), _stream())
    return out


def mc33_batch(vols, level, cap_v, cap_f):
    """mc33 of every volume of a (B,n0,n1,n2) batch at one level in one set of launches
    -> verts_vox [B][cap_v][3], faces [B][cap_f][3], normals, values [B][cap_v], counts (device int64 [B][2] = V, F)"""
    _chk(vols, torch.float32, "vols")
    B, n0, n1, n2 = vols.shape
    dev = vols.device
    nbytes = _lib.load().gn_mc33_batch_workspace_bytes(B, n0, n1, n2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    verts = torch.zeros((B, cap_v, 3), dtype=torch.float32, device=dev)      # rows past a volume's vertex count stay finite (padded consumers)
    faces = torch.empty((B, cap_f, 3), dtype=_i32, device=dev)
    normals = torch.empty((B, cap_v, 3), dtype=torch.float32, device=dev)
    values = torch.empty((B, cap_v), dtype=torch.float32, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int64, device=dev)
    _lib.call("gn_mc33_batch", _p(vols), B, n0, n1, n2, float(level), _p(ws), nbytes, _p(verts), _p(faces), _p(normals), _p(values), cap_v, cap_f,
              _p(counts), _stream())
    return verts, faces, normals, values, counts


def mc33_batch_profiled(vols, level, cap_v=None):
    """bench.py's hbm_members: one gn_mc33_batch of the (B,Q,Q,Q) batch with its stages timed inside the call (it synchronises) ->
    {stage: (ms, algorithmic bytes)}; bytes: classify = the volume once + one count byte per cell; scan = the block sums; vertices =
    (12 + 12 + 4) B per vertex + the count bytes; faces = 12 B per face + the count bytes"""
    _chk(vols, torch.float32, "vols")
    B, n0, n1, n2 = vols.shape
    dev = vols.device
    cap_v = int(cap_v or max(4096, 6 * n0 * n0))
    cap_f = 2 * cap_v + 64
    while True:
        nbytes = _lib.load().gn_mc33_batch_workspace_bytes(B, n0, n1, n2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        verts = torch.empty((B, cap_v, 3), dtype=torch.float32, device=dev)
        faces = torch.empty((B, cap_f, 3), dtype=_i32, device=dev)
        normals = torch.empty((B, cap_v, 3), dtype=torch.float32, device=dev)
        values = torch.empty((B, cap_v), dtype=torch.float32, device=dev)
        counts = torch.empty((B, 2), dtype=torch.int64, device=dev)
        ms = (ctypes.c_float * 4)()
        _lib.call("gn_mc33_batch_profiled", _p(vols), B, n0, n1, n2, float(level), _p(ws), nbytes, _p(verts), _p(faces), _p(normals), _p(values), cap_v, cap_f,
                  _p(counts), _stream(), ms)
        c = counts.cpu()
        nv, nf = int(c[:, 0].max()), int(c[:, 1].max())
        if nv <= cap_v and nf <= cap_f:
            break
        cap_v = max(nv, (nf + 1) // 2) + 64
        cap_f = 2 * cap_v + 64
    V, F = int(c[:, 0].sum()), int(c[:, 1].sum())
    cells = float(B) * (n0 - 1) * (n1 - 1) * (n2 - 1)
    return {"mc_classify": (ms[0], vols.numel() * 4.0 + cells), "mc_scan": (ms[1], cells / 1024 * 16.0), "mc_vertices_attrs": (ms[2], V * 28.0 + cells),
            "mc_faces": (ms[3], F * 12.0 + cells), "_mesh": (0.0, float(V))}


def mc33(vol, level, cap_v, cap_f):
    """-> verts_vox [cap_v][3], faces [cap_f][3], normals, values, counts (device int64 [2] = V, F)"""
    _chk(vol, torch.float32, "vol")
    n0, n1, n2 = vol.shape
    dev = vol.device
    nbytes = _lib.load().gn_mc33_workspace_bytes(n0, n1, n2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    verts = torch.empty((cap_v, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((cap_f, 3), dtype=_i32, device=dev)
    normals = torch.empty((cap_v, 3), dtype=torch.float32, device=dev)
    values = torch.empty(cap_v, dtype=torch.float32, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    _lib.call("gn_mc33", _p(vol), n0, n1, n2, float(level), _p(ws), nbytes, _p(verts), _p(faces), _p(normals), _p(values), cap_v, cap_f,
              _p(counts), _stream())
    return verts, faces, normals, values, counts


def gather_nn(vol, verts_vox, spacing):
    nv = verts_vox.shape[0]
    out = torch.empty(nv, dtype=torch.float32, device=vol.device)
    _lib.call("gn_gather_nn", _p(vol), *vol.shape, _p(verts_vox), nv, float(spacing), _p(out), _stream())
    return out


def gather_nn_batch(vols, verts_vox, spacing):
    """vols (B,n0,n1,n2), verts_vox (B,M,3) padded rows -> (B,M): the nearest-voxel value of every row in its own volume"""
    B, M = verts_vox.shape[:2]
    out = torch.empty((B, M), dtype=torch.float32, device=vols.device)
    _lib.call("gn_gather_nn_batch", _p(_chk(vols, torch.float32, "vols")), B, *vols.shape[1:], _p(_chk(verts_vox, torch.float32, "verts_vox")), M, float(spacing),
              _p(out), _stream())
    return out


def scale_verts(verts_vox, spacing):
    out = torch.empty_like(verts_vox)
    _lib.call("gn_scale_verts", _p(verts_vox), verts_vox.shape[0], float(spacing), _p(out), _stream())
    return out


def mesh_compact(verts, faces, on_surface):
    """delete_invalid_verts on the GPU -> (verts' (V',3) same dtype, faces' (F',3) int32); one host synchronisation for the two sizes.
    A face index outside [0, V) raises IndexError, as the reference's numpy indexing does (common/marching_cubes_util.py:41)."""
    assert verts.dim() == 2 and verts.shape[1] == 3 and verts.dtype in (torch.float32, torch.float64)
    verts = verts.contiguous()
    if faces.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"faces: expected an integer tensor, got {faces.dtype}")
    if verts.shape[0] >= 2 ** 31:
        raise ValueError("mesh_compact: more than 2^31 vertices")
    if faces.dtype == torch.int64:          # int32 on the device; an index that does not fit is out of range for any V < 2^31: keep it so
        faces = faces.clamp(min=-1, max=2 ** 31 - 1)
    faces = _chk(faces.to(torch.int32).contiguous(), torch.int32, "faces")
    flag = on_surface.to(torch.uint8).contiguous()
    V, F = verts.shape[0], faces.shape[0]
    if flag.numel() != V:
        raise ValueError("is_vert_on_surface must have one entry per vertex")
    nbytes = _lib.load().gn_mesh_compact_workspace_bytes(V, F)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=verts.device)
    out_v, out_f = torch.empty_like(verts), torch.empty_like(faces)
    counts = torch.empty(3, dtype=torch.int64, device=verts.device)
    _lib.call("gn_mesh_compact", _p(verts), verts.element_size() * 3, _p(faces), _p(flag), V, F, _p(ws), nbytes, _p(out_v), _p(out_f), _p(counts), _stream())
    nv, nf, bad = [int(c) for c in counts.cpu()]
    if bad:
        raise IndexError(f"delete_invalid_verts: a face index is out of bounds for {V} vertices")
    return out_v[:nv], out_f[:nf]


def mesh_largest_component(faces, num_verts, with_labels=False):
    """eval.py:497-503: -> (is_cc_vert bool [V], info dict) -- the vertices of the largest connected component of the mesh (ties: the
    component holding the lowest vertex index, as np.argmax over libigl's numbering); info: num_components, size, label [, labels]."""
    if faces.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"faces: expected an integer tensor, got {faces.dtype}")
    V, F = int(num_verts), faces.shape[0]
    if V >= 2 ** 31:
        raise ValueError("mesh_largest_component: more than 2^31 vertices")
    if F == 0:
        raise ValueError("attempt to get argmax of an empty sequence")          # what np.argmax(cc_sizes) says for a mesh without faces
    if faces.dtype == torch.int64:
        faces = faces.clamp(min=-1, max=2 ** 31 - 1)
    faces = _chk(faces.to(torch.int32).contiguous(), torch.int32, "faces")
    dev = faces.device
    nbytes = _lib.load().gn_mesh_largest_component_workspace_bytes(V)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
    mask = torch.empty(V, dtype=torch.uint8, device=dev)
    label = torch.empty(V, dtype=_i32, device=dev) if with_labels else None
    info = torch.empty(4, dtype=torch.int64, device=dev)
    _lib.call("gn_mesh_largest_component", _p(faces), F, V, _p(ws), nbytes, _p(mask), _p(label), _p(info), _stream())
    ncomp, size, lab, bad = [int(c) for c in info.cpu()]
    if bad:
        raise IndexError(f"connected components: a face index is out of bounds for {V} vertices")
    out = dict(num_components=ncomp, size=size, label=lab)
    if with_labels:
        out["labels"] = label
    return mask.bool(), out


# ------------------------------------------------------------------------------------------------ evaluation helpers
def nearest_neighbor(query, ref):
    """-> (idx int32 [Nq], d2 float32 [Nq]) exact 1-NN of every query point in `ref`"""
    query = _chk(query.float().contiguous(), torch.float32, "query")
    ref = _chk(ref.float().contiguous(), torch.float32, "ref")
    nq = query.shape[0]
    idx = torch.empty(nq, dtype=_i32, device=query.device)
    d2 = torch.empty(nq, dtype=torch.float32, device=query.device)
    _lib.call("gn_nearest_neighbor", _p(query), nq, _p(ref), ref.shape[0], _p(idx), _p(d2), _stream())
    return idx, d2
