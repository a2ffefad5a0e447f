"""dev-only A/B of one library build on the polyphase partial kernel (decoder shapes of the UNet at B=16); digests must agree between builds"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
dev = 'cuda'
def run(B, Dc, C1, C0, Cout, reps=5):
    g = torch.Generator().manual_seed(C1 + Cout + Dc)
    x = torch.randn(B, Dc, Dc, Dc, C1, generator=g).to(dev)
    a = (torch.rand(B, C1, generator=g) + 0.5).to(dev); d = (torch.randn(B, C1, generator=g) * 0.1).to(dev)
    inv = torch.full((B,), 0.5, device=dev)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) * 0.05
    w0, wm, _ = ops.polyphase_weights(w, C0)
    pkm = ops.pack_upconv_weight(wm, Cout, 4).to(dev)
    f = lambda: ops.upconv_partial(x, a, d, pkm, Cout, act_inv=inv)
    y = f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'B={B} coarse {Dc}^3 {C1}->8x{Cout}: {ms:.3f} ms {2.0*64*C1*Cout*B*Dc**3/ms/1e9:.1f} TF(eq) digest {dig}', flush=True)
for _ in range(2):
    run(16, 64, 64, 32, 32)      # dec2
    run(16, 32, 128, 64, 64)     # dec1
    run(16, 16, 256, 128, 128)   # dec0
