"""is the split conv's speed data-dependent (DVFS)?  same kernel, activations of different magnitude / with and without the per-sample scale"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
B, G, C0, Cout = 4, 128, 128, 128
w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
wps = ops.pack_conv_weight_split(w, 4).to(dev)
fl = 54.0 * C0 * Cout * B * G ** 3
def t(f, reps=5):
    f(); f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
x = torch.randn(B, G, G, G, C0, device=dev)
d = torch.zeros(B, C0, device=dev)
ones = torch.ones(B, device=dev)
for rep in range(2):
    for mag in (1.0, 2.0 ** -4, 2.0 ** -8, 2.0 ** -12, 0.0):
        a = torch.full((B, C0), mag, device=dev)
        ms = t(lambda: ops.conv3d_gcr_split(x, None, a, d, wps, Cout))
        ms2 = t(lambda: ops.conv3d_gcr_split(x, None, a, d, wps, Cout, act_inv=ones))
        print(f'|y|~{mag:g}: {ms:.2f} ms {fl/ms/1e9:.1f} TF(eq)   with act_inv: {ms2:.2f} ms {fl/ms2/1e9:.1f}')
# sparse volume like the benchmark's: constant per channel
xs = torch.zeros(B, G, G, G, C0, device=dev)
for dv in (0.1, 1.5):
    dd = torch.full((B, C0), dv, device=dev)
    a = torch.ones(B, C0, device=dev)
    ms = t(lambda: ops.conv3d_gcr_split(xs, None, a, dd, wps, Cout))
    print(f'constant volume d={dv}: {ms:.2f} ms {fl/ms/1e9:.1f} TF(eq)')
