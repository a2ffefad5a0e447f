"""dev-only A/B of one library build on the lattice (brick) sampler: the bench's chunks (Q = 128 over 128^3 x 32 and over 32^3 x 32, 2^20 rows per launch), Q = 256; digests must agree"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(3)
for G, Q, C in ((128, 128, 32), (32, 128, 32), (128, 256, 32), (16, 24, 64), (128, 128, 128)):
    vol = torch.randn(G, G, G, C, generator=g).to(dev)
    slab = Q * Q
    M = min(Q ** 3, max(slab, (2 ** 20 // slab) * slab))
    out = ops.trilinear_sample(vol, Q=Q, m0=0, M=M); torch.cuda.synchronize()
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    last = ops.trilinear_sample(vol, Q=Q, m0=Q ** 3 - M, M=M)           # the chunk that touches the upper faces
    h2 = hashlib.sha1(last.cpu().numpy().tobytes()).hexdigest()[:12]
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [ops.trilinear_sample(vol, Q=Q, m0=0, M=M, out=out) for _ in range(50)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f'G={G} Q={Q} C={C} M={M}: {ms * 1e3:.1f} us  {M * C * 4 / ms / 1e6:.0f} GB/s written  digests {h} {h2}', flush=True)
