"""dev-only: time the 128 -> 128 conv at 128^3 (B = 4) on a scattered operand (exact zeros: the affine-in-weights regime) and on N(0,1), for the
library named by GARMENTNETS_HIP_LIB (timing-only ablation builds of conv3d_split_wide_kernel)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
B, G, C = 4, 128, 128
w = torch.randn(C, C, 3, 3, 3) * 0.02
pk = ops.pack_conv_weight_split(w, 4).to('cuda')
a = torch.ones(B, C, device='cuda'); d = torch.zeros(B, C, device='cuda')
g = torch.Generator(device='cuda').manual_seed(0)
out = []
for name, frac in (("scattered", 0.0024), ("N(0,1)", 1.0)):
    x = torch.randn(B, G, G, G, C, device='cuda', generator=g)
    if frac < 1.0:
        x = x * (torch.rand(B, G, G, G, 1, device='cuda', generator=g) < frac)
    ops.conv3d_gcr_split(x, None, a, d, pk, C); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(10): ops.conv3d_gcr_split(x, None, a, d, pk, C)
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out.append(f"{name} {ms:.3f} ms {54.0*C*C*B*G**3/ms/1e9:.1f} TF-eq")
    del x
    time.sleep(0.5)
print(os.environ.get("GARMENTNETS_HIP_LIB", "product"), " | ".join(out), flush=True)
