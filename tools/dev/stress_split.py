"""dev-only: repeated split-operand conv / decoder launches against the fp32 kernels (a race in the counted-wait DMA pipelines would
show up as an occasional large difference)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
torch.manual_seed(0)
shapes = [(128, 0, 128, (32, 32, 64), 2), (32, 64, 32, (32, 32, 32), 2), (64, 0, 64, (32, 32, 32), 2), (128, 256, 128, (16, 16, 32), 4), (128, 0, 32, (64, 64, 64), 1), (16, 0, 32, (9, 17, 33), 3)]
worst = 0.0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    C0, C1, Cout, (D, H, W), B = shapes[it % len(shapes)]
    x0 = torch.randn(B, D, H, W, C0, device=dev)
    x1 = torch.randn(B, D // 2, H // 2, W // 2, C1, device=dev) if C1 and D % 2 == 0 and H % 2 == 0 and W % 2 == 0 else None
    cin = C0 + (C1 if x1 is not None else 0)
    a = torch.rand(B, cin, device=dev) + 0.5; d = torch.randn(B, cin, device=dev)
    w = torch.randn(Cout, cin, 3, 3, 3) / (27 * cin) ** 0.5
    ref = ops.conv3d_gcr(x0, x1, a, d, ops.pack_conv_weight(w).to(dev), Cout)
    pk = ops.pack_conv_weight_split(w, 4).to(dev)
    for rep in range(3):
        out = ops.conv3d_gcr_split(x0, x1, a, d, pk, Cout)
        e = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        worst = max(worst, e)
        assert e < 2e-5, (it, rep, C0, C1, Cout, D, H, W, e)
g = torch.Generator().manual_seed(1)
for k0 in (32, 128):
    dims = [k0, 256, 256, 1]
    raw = [(torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5, torch.randn(dims[i + 1], generator=g) * 0.1, torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1) for i in range(3)]
    pk = ops.pack_decode_split(raw).to(dev)
    layers = tuple((ops.pack_kpair(w).to(dev) if i < 2 else w.contiguous().to(dev), b.to(dev), sc.to(dev), sh.to(dev), dims[i + 1]) for i, (w, b, sc, sh) in enumerate(raw))
    for M in (262144, 100001, 333):
        xin = ops.new_rows(M, k0, dev); xin.copy_(torch.randn(M, k0, device=dev))
        ref = ops.implicit_decode(None, layers, M=M, xin=xin)
        for rep in range(10):
            out = ops.implicit_decode_split(xin, pk)
            e = (out - ref).abs().max().item()
            worst = max(worst, e)
            assert e < 3e-5, (k0, M, rep, e)
print("stress ok, worst relative / absolute difference", worst)
