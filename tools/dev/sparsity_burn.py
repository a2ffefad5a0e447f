"""dev-only: what operand sparsity is worth under the power cap -- the 128 -> 128 conv (dominant kernel) on inputs with different fractions of exact
zeros reaching the matrix cores (GroupNorm shift d = 0 keeps zeros zero; d != 0 is today's form: every voxel becomes non-zero)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
B, G, C = 4, 128, 128
w = torch.randn(C, C, 3, 3, 3) * 0.02
pk = ops.pack_conv_weight_split(w, 4).to('cuda')
a = torch.ones(B, C, device='cuda')
g = torch.Generator(device='cuda').manual_seed(0)
def vol(frac_nonzero_voxels, relu_like):
    x = torch.randn(B, G, G, G, C, device='cuda', generator=g)
    if relu_like:
        x = torch.relu(x)                                 # ~50 % exact zeros, element-wise
    if frac_nonzero_voxels < 1.0:
        m = torch.rand(B, G, G, G, 1, device='cuda', generator=g) < frac_nonzero_voxels
        x = x * m
    return x
for name, x, dshift in (("dense N(0,1), d=0", vol(1.0, False), 0.0), ("post-ReLU (50 % zeros), d=0", vol(1.0, True), 0.0), ("post-ReLU, d=-0.5 (today: GN shift in the operand)", vol(1.0, True), -0.5),
                        ("scattered 0.24 % voxels, d=0", vol(0.0024, False), 0.0), ("scattered 0.24 % voxels, d=0.3 (today)", vol(0.0024, False), 0.3)):
    d = torch.full((B, C), dshift, device='cuda')
    ops.conv3d_gcr_split(x, None, a, d, pk, C); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(10): ops.conv3d_gcr_split(x, None, a, d, pk, C)
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms:.3f} ms, {54.0*C*C*B*G**3/ms/1e9:.1f} TFLOP/s-eq", flush=True)
    del x
    time.sleep(1.0)
