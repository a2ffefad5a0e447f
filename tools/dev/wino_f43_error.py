"""round 6: error growth of Winograd F(4,3) along x against F(2,3) and the direct form, all in the f16x2 arithmetic of csrc/unet_wino.hip
(fp32 input transform BEFORE the exact two-plane fp16 split, fp64 weight transform on the pack side with a per-output-channel power-of-two scale, three fp16
products per fp32 product accumulated in fp32, fp32 output transform), on the first encoder convolution's shape: 128 input channels x 9 (dz, dy) taps = 1152
one-dimensional 3-tap convolutions summed per output.  CPU only (numpy); yardstick: the same sum in fp64.  usage: python tools/dev/wino_f43_error.py"""
import numpy as np

rng = np.random.default_rng(0)
K, N, X = 128 * 9, 32, 512          # summed rows, output channels, outputs along x (multiple of 4)


def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def prod3(a, w):
    """sum over K of a[K, P] * w[K, N] in f16x2: (a_lo w_hi + a_hi w_lo) + a_hi w_hi, every product exact in fp32, fp32 accumulation"""
    ah, al = split(a)
    wh, wl = split(w)
    return (al.T @ wh + ah.T @ wl) + ah.T @ wh


def pack(u):
    """per-output-channel power-of-two scale: row maximum in [1, 2) (ops.pack_conv_weight_split_wino); u [..., N] fp64 -> (fp32 scaled, 1/scale)"""
    m = np.abs(u).reshape(-1, u.shape[-1]).max(axis=0)
    s = 2.0 ** -np.floor(np.log2(m))
    return (u * s).astype(np.float32), (1.0 / s).astype(np.float32)


F23 = dict(BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float),
           G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float),
           AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float), m=2)
F43 = dict(BT=np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float),
           G=np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float),
           AT=np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float), m=4)


def run(scattered):
    d = rng.standard_normal((K, X + 2))
    if scattered:
        d *= rng.random((1, X + 2)) < 0.05
    d = d.astype(np.float32)
    g = (rng.standard_normal((K, 3, N)) / np.sqrt(3 * K))
    ref = sum(d[:, t:t + X].astype(np.float64).T @ g[:, t] for t in range(3))            # [X, N] fp64
    out = {}
    # direct form: three taps, each an f16x2 product sum
    wd, isc = pack(np.concatenate([g[:, 0], g[:, 1], g[:, 2]], axis=0))
    a = np.concatenate([d[:, t:t + X] for t in range(3)], axis=0)
    out["direct"] = prod3(a, wd) * isc
    for name, F in (("F(2,3)", F23), ("F(4,3)", F43)):
        m, a_ = F["m"], F["BT"].shape[0]
        tiles = X // m
        dt = np.stack([d[:, m * i:m * i + a_] for i in range(tiles)], axis=1)                # [K, tiles, a]
        v = np.einsum("ja,kta->kjt", F["BT"].astype(np.float32), dt).astype(np.float32)      # input transform in fp32
        u = np.einsum("jt,ktn->kjn", F["G"], g)                                              # weight transform in fp64
        us, isc = pack(u.reshape(-1, N))
        us = us.reshape(K, a_, N)
        mm = np.stack([prod3(v[:, j], us[:, j]) for j in range(a_)], axis=0)                  # [a, tiles, N] fp32 accumulators
        y = np.einsum("oj,jtn->ton", F["AT"].astype(np.float32), mm).astype(np.float32)      # output transform in fp32
        out[name] = y.reshape(X, N) * isc
    scale = np.abs(ref).max()
    return {k: float(np.abs(v - ref).max()) for k, v in out.items()}, scale


for scattered in (False, True):
    errs, scale = [], 0
    for rep in range(4):
        e, sc = run(scattered)
        errs.append(e)
        scale = max(scale, sc)
    print(("scattered (5 % occupied)" if scattered else "dense N(0,1)        "), f"|out| max {scale:.2f}  max abs error vs fp64 over 4 draws:",
          {k: f"{max(e[k] for e in errs):.2e}" for k in errs[0]})
