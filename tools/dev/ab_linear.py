"""dev-only A/B of one library build on gn_linear at the PointNet++ layer shapes of a 16-garment step; digests must agree"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(2)
tot = 0.0
for M, K, N in ((96000, 131, 128), (96000, 128, 128), (96000, 128, 192), (48000, 384, 256), (48000, 256, 128), (12000, 1280, 256), (12000, 256, 256), (6000, 131, 128), (6000, 128, 64), (96000, 6, 32)):
    x = ops.new_rows(M, K, dev); x.copy_(torch.randn(M, K, generator=g).to(dev))
    w = torch.randn(N, K, generator=g).to(dev) * K ** -0.5
    b = torch.randn(N, generator=g).to(dev)
    sc = (torch.rand(N, generator=g) + 0.5).to(dev); sh = torch.randn(N, generator=g).to(dev)
    wp = torch.zeros(N, ops.pad4(K), device=dev); wp[:, :K] = w
    y = ops.linear(x, wp, b, sc, sh, relu=True, K=K)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.linear(x, wp, b, sc, sh, relu=True, K=K, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20; tot += ms
    print(f'M={M} K={K} N={N}: {ms * 1e3:.1f} us  {2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s  digest {hashlib.sha1(y.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]}', flush=True)
print(f'sum {tot * 1e3:.1f} us')
