"""dev: fixed (prologue + epilogue) vs per-slice time of the split convs: time against the number of 16-channel input slices"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
def t(f, reps=6):
    f(); f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for Cout in (128, 32):
    B, G = 4, 128
    res = []
    for C0 in (16, 32, 64, 128, 256):
        w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
        wps = ops.pack_conv_weight_split(w, 4).to(dev)
        x = torch.randn(B, G, G, G, C0, device=dev)
        a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
        ms = min(t(lambda: ops.conv3d_gcr_split(x, None, a, d, wps, Cout)) for _ in range(3))
        res.append((C0 // 16, ms))
        print(f'Cout={Cout} slices={C0//16}: {ms:.3f} ms', flush=True)
        del x
    (s0, t0), (s1, t1) = res[0], res[-1]
    per = (t1 - t0) / (s1 - s0)
    print(f'Cout={Cout}: per slice {per:.3f} ms, fixed {t0 - per * s0:.3f} ms  (128-channel layer: fixed share {(t0 - per*s0) / (8*per + t0 - per*s0):.3f})')
