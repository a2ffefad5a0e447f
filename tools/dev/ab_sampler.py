"""dev-only: lattice sampler, brick kernel (chunks of whole slabs) vs per-query kernel (chunk shifted by one row)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Q = 128
vol = torch.randn(G, G, G, 128, device='cuda')
M = 16 * Q * Q
out = ops.new_rows(M, 128, 'cuda')
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for name, m0 in (("brick", 16 * Q * Q), ("per-query", 16 * Q * Q + 1)):
    ms = t(lambda: ops.trilinear_sample(vol, Q=Q, m0=m0, M=M - (m0 & 1), out=out[:M - (m0 & 1)]))
    print(f"G={G} {name:10s}: {ms:.4f} ms per {M} rows  ({M*512/ms/1e6:.0f} GB/s of output)")
