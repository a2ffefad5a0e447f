"""dev-only: time the split-operand decoder MLP against the fp32 one on a 128^3 lattice worth of rows"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
M = 1 << 18
out_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K0 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(0)
dims = [K0, 256, 256, out_ch]
raw = []
for i in range(3):
    raw.append((torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5, torch.randn(dims[i + 1], generator=g) * 0.1,
                torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1))
xin = ops.new_rows(M, K0, dev); xin.copy_(torch.randn(M, K0, generator=g).to(dev))
pk = ops.pack_decode_split(raw).to(dev)
layers = tuple((ops.pack_kpair(w).to(dev) if i < 2 else w.contiguous().to(dev), b.to(dev), sc.to(dev), sh.to(dev), dims[i + 1]) for i, (w, b, sc, sh) in enumerate(raw))
fl = 2.0 * M * (K0 * 256 + 256 * 256 + 256 * out_ch)
def t(f, reps=8):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
a = ops.implicit_decode_split(xin, pk); b = ops.implicit_decode(None, layers, M=M, xin=xin)
print('max diff split vs fp32', (a - b).abs().max().item())
ms = t(lambda: ops.implicit_decode(None, layers, M=M, xin=xin)); print(f'fp32 : {ms:.3f} ms per {M} rows  {fl/ms/1e9:.1f} TF')
ms = t(lambda: ops.implicit_decode_split(xin, pk)); print(f'split: {ms:.3f} ms per {M} rows  {fl/ms/1e9:.1f} TF(eq)')
