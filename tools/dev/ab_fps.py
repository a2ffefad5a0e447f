"""dev-only A/B of one library build on farthest-point sampling (the bench step's two levels + the batch-of-one latency); digests must agree"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(11)
for B, n, ratio in ((16, 6000, 0.5), (16, 3000, 0.25), (1, 6000, 0.5), (1, 3000, 0.25), (16, 10000, 0.5), (4, 777, 0.5), (8, 8000, 0.25), (8, 12288, 0.25), (8, 16384, 0.125), (3, 100, 0.5), (2, 6144, 0.5), (2, 6145, 0.5)):
    pos = (torch.rand(B * n, 3, generator=g) - 0.5).to(dev)
    pos[5] = pos[3]                                     # an exact duplicate: ties on the way
    m = ops.fps_count(n, ratio)
    ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    optr = torch.arange(0, (B + 1) * m, m, dtype=torch.int32, device=dev)
    idx = ops.fps(pos, ptr, optr, n, B * m); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [ops.fps(pos, ptr, optr, n, B * m) for _ in range(10)]; e1.record(); torch.cuda.synchronize()
    print(f'B={B} n={n} m={m}: {e0.elapsed_time(e1) / 10:.4f} ms  {e0.elapsed_time(e1) / 10 / (m - 1) * 1e3:.3f} us/step  digest {hashlib.sha1(idx.cpu().numpy().tobytes()).hexdigest()[:12]}', flush=True)
