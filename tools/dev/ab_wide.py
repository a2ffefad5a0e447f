"""A/B of the wide split conv (128->128 @128^3, B=4) under GARMENTNETS_CONV_EXP settings: run as separate processes"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
def t(f, reps=6):
    f(); f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for (B, G, C0, Cout) in ((4, 128, 128, 128), (4, 128, 128, 32), (4, 128, 32, 32)):
    w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
    wps = ops.pack_conv_weight_split(w, 4).to(dev)
    x = torch.randn(B, G, G, G, C0, device=dev)
    a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
    fl = 54.0 * C0 * Cout * B * G ** 3
    ms = min(t(lambda: ops.conv3d_gcr_split(x, None, a, d, wps, Cout)) for _ in range(3))
    print(f'EXP={os.environ.get("GARMENTNETS_CONV_EXP","0")} {C0}->{Cout}: {ms:.2f} ms {fl/ms/1e9:.1f} TF(eq)')
