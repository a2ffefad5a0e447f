"""dev-only A/B of one library build on the Gaussian gradient magnitude (the step's 16 x 128^3 and config[4]'s 8 x 256^3); digests must agree across builds"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
g = torch.Generator().manual_seed(3)
for B, Q in ((16, 128), (8, 256), (1, 128), (3, 50)):
    vols = (torch.rand(B, Q, Q, Q, generator=g) * 2 - 0.5).cuda()
    for bits in (64, 32):
        for rng in (True, False):
            if not rng and bits == 32:
                continue
            f = (lambda: ops.ggm3d_batch_range(vols, 0.5, bits)[0]) if rng else (lambda: ops.ggm3d_batch(vols, 0.5))
            out = f(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); [f() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
            print(f"B={B} Q={Q} fp{bits} range={int(rng)}: {e0.elapsed_time(e1) / 10:.4f} ms  digest {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
