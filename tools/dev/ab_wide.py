"""dev-only A/B of one library build on the 128-wide conv shapes (zero and N(0,1) operands); digests must agree between builds"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
dev = 'cuda'
def run(B, G, C0, Cout, zero, reps=4):
    g = torch.Generator().manual_seed(C0 + Cout + G)
    x = torch.zeros(B, G, G, G, C0, device=dev) if zero else torch.randn(B, G, G, G, C0, generator=g).to(dev)
    a = (torch.rand(B, C0, generator=g) + 0.5).to(dev); d = torch.zeros(B, C0, device=dev) if zero else (torch.randn(B, C0, generator=g) * 0.1).to(dev)
    inv = torch.full((B,), 0.5, device=dev)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) * 0.05
    pk = ops.pack_conv_weight_split(w, 4).to(dev)
    f = lambda: ops.conv3d_gcr_split(x, None, a, d, pk, Cout, relu=True, with_stats=True, act_inv=inv)
    y, (s, q, V) = f(); torch.cuda.synchronize()
    kern = _lib.load().gn_last_kernel().decode()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'B={B} {G}^3 {C0}->{Cout} {"zeros" if zero else "randn"}: {ms:.3f} ms {54.0*C0*Cout*B*G**3/ms/1e9:.1f} TF(eq) digest {dig} [{kern}]', flush=True)
for rep in range(2):
    run(8, 128, 128, 128, True)
    run(8, 128, 128, 128, False)
    run(16, 32, 128, 128, False, reps=10)
    run(16, 32, 384, 128, False, reps=10)
