"""round 6: the 32-wide Winograd kernel (csrc/unet_wino32.hip) against the x-strip kernel on the UNet's 32- / 64-wide shapes: error vs fp64 on small volumes,
time / TF-eq / digest at full size.  usage: python tools/dev/ab_wino32.py [check|time|all]"""
import hashlib
import sys
import time

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops

DEV = "cuda"


def check(B, dims, C0, Cout, scattered, with_partial=False):
    g = torch.Generator().manual_seed(C0 + Cout + dims[2])
    D, H, W = dims
    x = torch.randn(B, C0, D, H, W, generator=g)
    if scattered:
        x = x * (torch.rand(B, 1, D, H, W, generator=g) < 0.05)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    gamma, beta = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    pre = F.conv3d(F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1)
    part = None
    if with_partial:
        part = torch.randn(B, D // 2, H // 2, W // 2, 8 * Cout, generator=g)
        pp = part.double().view(B, D // 2, H // 2, W // 2, 2, 2, 2, Cout).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, Cout, D, H, W)
        pre = pre + pp
        part = part.to(DEV)
    ref = F.relu(pre)
    s0 = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    st = ops.channel_stats(s0)
    a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV), with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    cl = lambda t: t.permute(0, 4, 1, 2, 3).cpu().double()
    err = lambda t: float((cl(t) - ref).abs().max())
    e32 = err(ops.conv3d_gcr(s0, None, a0, d0, ops.pack_conv_weight(w).to(DEV), Cout)) if not with_partial else float("nan")
    e_dir = err(ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), Cout, act_inv=inv, partial=part))
    yw, (sm, sq, V) = ops.conv3d_gcr_split_wino(s0, a, d, ops.pack_conv_weight_split_wino(w).to(DEV), Cout, act_inv=inv, with_stats=True, partial=part)
    kern = ops._lib.load().gn_last_kernel().decode()
    prep = ops.conv_affine_pack(w.to(DEV).contiguous(), a0, d0, st, wino=True)
    yr = ops.conv3d_gcr_split_persample(s0, prep, partial=part)
    es = float((sm.cpu() - yw.double().sum(dim=(1, 2, 3)).cpu()).abs().max()) / max(1.0, float(sm.abs().max()))
    print(f"B={B} {dims} {C0}->{Cout} {'scattered' if scattered else 'dense'}{' +partial' if with_partial else ''}: {kern}: err vs fp64: fp32-MFMA {e32:.2e}, "
          f"strip {e_dir:.2e}, wino32 literal {err(yw):.2e}, affine-in-weights {err(yr):.2e}; stats rel err {es:.1e}; rerun equal {torch.equal(yw, ops.conv3d_gcr_split_wino(s0, a, d, ops.pack_conv_weight_split_wino(w).to(DEV), Cout, act_inv=inv, partial=part))}",
          flush=True)


def timeit(B, G, C0, Cout, zeros=False, with_partial=False, reps=5):
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(B, G, G, G, C0) if zeros else torch.randn(B, G, G, G, C0, generator=g)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    s0 = x.to(DEV)
    a = torch.ones(B, C0, device=DEV)
    d = torch.zeros(B, C0, device=DEV)
    part = torch.randn(B, G // 2, G // 2, G // 2, 8 * Cout, generator=g).to(DEV) if with_partial else None
    pk_d = ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV)
    pk_w = ops.pack_conv_weight_split_wino(w).to(DEV)
    flops = 54.0 * C0 * Cout * B * G ** 3
    res = {}
    for name, fn in (("strip", lambda: ops.conv3d_gcr_split(s0, None, a, d, pk_d, Cout, with_stats=True, partial=part)),
                     ("wino32", lambda: ops.conv3d_gcr_split_wino(s0, a, d, pk_w, Cout, with_stats=True, partial=part))):
        y = fn()[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = fn()[0]
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = (ms, y)
        print(f"  {name:7s} {ops._lib.load().gn_last_kernel().decode():36s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF-eq  sha {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
    dy = float((res["strip"][1] - res["wino32"][1]).abs().max())
    print(f"B={B} G={G} {C0}->{Cout} {'zeros' if zeros else 'N(0,1)'}{' +partial' if with_partial else ''}: speedup {res['strip'][0] / res['wino32'][0]:.3f}x, max |strip - wino32| {dy:.2e}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("check", "all"):
        for cfg in [(2, (8, 16, 16), 32, 32, False), (2, (8, 16, 16), 32, 32, True), (1, (8, 8, 8), 128, 32, False), (2, (16, 8, 24), 64, 64, False),
                    (1, (16, 16, 16), 128, 32, True), (1, (8, 8, 40), 16, 32, False), (3, (24, 16, 8), 32, 96, False)]:
            check(*cfg)
        check(2, (8, 16, 16), 32, 32, False, with_partial=True)
        check(1, (16, 16, 16), 64, 64, False, with_partial=True)
    if what == "prof":                     # under rocprofv3 --pmc: one launch of each kernel on the 128 -> 32 layer (zeros and N(0,1)) and the 32 -> 32 layer
        for B, G, C0, Cout, z in ((16, 128, 128, 32, True),):
            g = torch.Generator().manual_seed(1)
            x = (torch.zeros(B, G, G, G, C0) if z else torch.randn(B, G, G, G, C0, generator=g)).to(DEV)
            w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
            a, d = torch.ones(B, C0, device=DEV), torch.zeros(B, C0, device=DEV)
            pk, pkd = ops.pack_conv_weight_split_wino(w).to(DEV), ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV)
            for _ in range(2):
                ops.conv3d_gcr_split_wino(x, a, d, pk, Cout, with_stats=True)
                ops.conv3d_gcr_split(x, None, a, d, pkd, Cout, with_stats=True)
            torch.cuda.synchronize()
    if what == "w32":                      # the new kernel alone (ablation builds: GARMENTNETS_HIP_LIB=tools/dev/_build/lib_<name>.so)
        for B, G, C0, Cout, z in ((16, 128, 128, 32, False), (16, 128, 128, 32, True), (16, 128, 32, 32, False), (16, 64, 64, 64, False)):
            g = torch.Generator().manual_seed(1)
            x = (torch.zeros(B, G, G, G, C0) if z else torch.randn(B, G, G, G, C0, generator=g)).to(DEV)
            w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
            a, d, pk = torch.ones(B, C0, device=DEV), torch.zeros(B, C0, device=DEV), ops.pack_conv_weight_split_wino(w).to(DEV)
            fn = lambda: ops.conv3d_gcr_split_wino(x, a, d, pk, Cout, with_stats=True)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"   B={B} G={G} {C0}->{Cout} {'zeros ' if z else 'N(0,1)'} {ms:8.3f} ms {54.0 * C0 * Cout * B * G ** 3 / ms / 1e9:7.1f} TF-eq", flush=True)
    if what in ("time", "all"):
        timeit(16, 128, 128, 32)
        timeit(16, 128, 128, 32, zeros=True)
        timeit(16, 128, 32, 32)
        timeit(16, 128, 32, 32, with_partial=True)
        timeit(16, 64, 32, 32)
        timeit(16, 64, 32, 64)
        timeit(16, 64, 64, 64)
