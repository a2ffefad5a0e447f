"""dev-only: the 32-wide conv shapes on an all-zero input (operand exactly zero: the not-power-bound regime of the at-rest layers) and on randn"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
dev = 'cuda'
def run(B, dims, C0, Cout, zero, reps=5):
    g = torch.Generator().manual_seed(1)
    D, H, W = dims
    x = torch.zeros(B, D, H, W, C0, device=dev) if zero else torch.randn(B, D, H, W, C0, generator=g).to(dev)
    a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) * 0.05
    pk = ops.pack_conv_weight_split(w, 4).to(dev)
    f = lambda: ops.conv3d_gcr_split(x, None, a, d, pk, Cout, relu=True, with_stats=True)
    f(); torch.cuda.synchronize()
    kern = _lib.load().gn_last_kernel().decode()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'B={B} {dims} {C0}->{Cout} {"zeros" if zero else "randn"}: {ms:.3f} ms {54.0*C0*Cout*B*D*H*W/ms/1e9:.1f} TF(eq) [{kern}]', flush=True)
for zero in (True, False):
    run(16, (128, 128, 128), 128, 32, zero, reps=4)
    run(16, (128, 128, 128), 32, 32, zero, reps=4)
    run(16, (64, 64, 64), 32, 32, zero, reps=8)
    run(16, (64, 64, 64), 64, 64, zero, reps=8)
