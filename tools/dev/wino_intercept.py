"""dev-only: fixed cost per launch of the Winograd conv kernel (prologue + epilogue per workgroup): time against the number of 16-channel slices at Cout = 128, 128^3, B = 4"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
B, G, Cout = 4, 128, 128
g = torch.Generator().manual_seed(0)
pts = []
for C in (16, 32, 64, 128, 256):
    x = (torch.randn(B, G, G, G, C, generator=g) * (torch.rand(B, G, G, G, 1, generator=g) < 0.0025)).to(dev)
    w = torch.randn(Cout, C, 3, 3, 3, generator=g) * 0.05
    st = ops.channel_stats(x)
    a0, d0 = ops.groupnorm_affine(st, None, 8 if C >= 8 else 1, 1e-5, torch.ones(C, device=dev), torch.zeros(C, device=dev))
    for name, wino in (("direct", False), ("wino", True)):
        prep = ops.conv_affine_pack(w.to(dev).contiguous(), a0, d0, st, wino=wino)
        f = lambda: ops.conv3d_gcr_split_persample(x, prep, with_stats=True)
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        pts.append((name, C, ms))
        print(f"{name} Cin={C}: {ms:.3f} ms", flush=True)
    del x
for name in ("direct", "wino"):
    p = [(c, m) for n, c, m in pts if n == name]
    (c1, m1), (c2, m2) = p[1], p[-2]
    slope = (m2 - m1) / (c2 - c1)
    print(f"{name}: slope {slope * 16:.3f} ms per 16-channel slice, intercept {m1 - slope * c1:.3f} ms ({(m1 - slope * c1) / p[-2][1] * 100:.1f} % of the Cin = 128 launch)")
