"""dev-only: fixed cost per workgroup of conv3d_split_wide_kernel -- time of the Cin -> 128 conv at 128^3 (B = 4, scattered operand) against Cin:
the intercept of the linear fit is what a workgroup spends outside its slice loop (launch, prologue, epilogue)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
B, G, Cout = 4, 128, 128
g = torch.Generator(device='cuda').manual_seed(0)
res = []
for C in (16, 32, 64, 128):
    w = torch.randn(Cout, C, 3, 3, 3) * 0.02
    pk = ops.pack_conv_weight_split(w, 4).to('cuda')
    a = torch.ones(B, C, device='cuda'); d = torch.zeros(B, C, device='cuda')
    x = torch.randn(B, G, G, G, C, device='cuda', generator=g) * (torch.rand(B, G, G, G, 1, device='cuda', generator=g) < 0.0024)
    ops.conv3d_gcr_split(x, None, a, d, pk, Cout); torch.cuda.synchronize()
    kern = _lib.load().gn_last_kernel().decode()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for _ in range(n): ops.conv3d_gcr_split(x, None, a, d, pk, Cout)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res.append((C, ms))
    print(f"Cin={C}: {ms:.3f} ms  {kern}", flush=True)
    del x
(c1, t1), (c2, t2) = res[1], res[3]
slope = (t2 - t1) / (c2 - c1)
print(f"slope {slope * 16:.4f} ms per 16-channel slice, intercept {t2 - slope * c2:.3f} ms of {t2:.3f} ms at Cin=128 ({(t2 - slope * c2) / t2 * 100:.1f} %)")
