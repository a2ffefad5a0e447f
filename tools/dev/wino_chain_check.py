"""dev-only: the chained Winograd conv kernel gives the same bits for every chain length (GARMENTNETS_WINO_CHAIN, read once per process):
prints digests of outputs / statistics for dense, per-sample-pack and occupancy-aware launches; run under several chain lengths and diff."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
dig = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
g = torch.Generator().manual_seed(7)
for (B, D, H, W, C, Cout) in ((3, 16, 32, 32, 32, 128), (2, 8, 24, 40, 64, 256), (5, 12, 16, 16, 16, 128)):
    x = torch.randn(B, D, H, W, C, generator=g).to(dev)
    w = torch.randn(Cout, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    st = ops.channel_stats(x)
    grp = 8 if C % 8 == 0 else 1
    a, d, inv = ops.groupnorm_affine(st, None, grp, 1e-5, gamma, beta, with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, grp, 1e-5, gamma, beta)
    y, (sm, sq, V) = ops.conv3d_gcr_split_wino(x, a, d, ops.pack_conv_weight_split_wino(w).to(dev), Cout, act_inv=inv, with_stats=True)
    prep = ops.conv_affine_pack(w.to(dev).contiguous(), a0, d0, st, wino=True)
    y2, (sm2, sq2, V2) = ops.conv3d_gcr_split_persample(x, prep, with_stats=True)
    ref_s = y.double().sum(dim=(1, 2, 3)); ref_q = (y.double() ** 2).sum(dim=(1, 2, 3))
    print(f"B={B} {D}x{H}x{W} {C}->{Cout}: literal {dig(y)} at-rest {dig(y2)} stats rel err {((sm - ref_s).abs().max() / ref_s.abs().max()).item():.1e} "
          f"{((sq - ref_q).abs().max() / ref_q.abs().max()).item():.1e} sums {sm.sum().item():.10e} {sq2.sum().item():.10e}", flush=True)
