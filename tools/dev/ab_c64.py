"""dev-only A/B of one library build on the 64-wide conv shapes of the UNet (run once per build with GARMENTNETS_HIP_LIB; digests must agree)"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
dev = 'cuda'
def run(B, dims, C0, Cout, reps=5, partial=False):
    g = torch.Generator().manual_seed(C0 * 7 + Cout + dims[0])
    D, H, W = dims
    x = (torch.randn(B, D, H, W, C0, generator=g)).to(dev)
    a = (torch.rand(B, C0, generator=g) + 0.5).to(dev); d = (torch.randn(B, C0, generator=g) * 0.1).to(dev)
    inv = torch.full((B,), 0.5, device=dev)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) * 0.05
    pk = ops.pack_conv_weight_split(w, 4).to(dev)
    part = (torch.randn(B, D // 2, H // 2, W // 2, 8 * Cout, generator=g)).to(dev) if partial else None
    f = lambda: ops.conv3d_gcr_split(x, None, a, d, pk, Cout, relu=True, with_stats=True, act_inv=inv, partial=part)
    y, (s, q, V) = f(); torch.cuda.synchronize()
    kern = _lib.load().gn_last_kernel().decode()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'B={B} {dims} {C0}->{Cout} partial={partial}: {ms:.3f} ms {54.0*C0*Cout*B*D*H*W/ms/1e9:.1f} TF(eq) digest {dig} sum {float(s.sum()):.6e} [{kern}]', flush=True)
for _ in range(2):
    run(16, (64, 64, 64), 32, 64)
    run(16, (64, 64, 64), 64, 64)
    run(16, (64, 64, 64), 64, 64, partial=True)
    run(16, (32, 32, 32), 64, 64)
    run(16, (128, 128, 128), 32, 32, reps=3)
    run(16, (128, 128, 128), 128, 32, reps=3)
