"""dev-only A/B of one library build on the folded scalar decoder [32,256,256,1] (the lattice decoder of the bench step); digests must agree"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(5)
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REST = len(sys.argv) > 2 and sys.argv[2] == 'rest'     # every row the same (zeros): the same instruction stream on operands that do not toggle
dims = [32, NH, NH, 1]
raw = []
for i in range(3):
    w = torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5 * (0.3 if i == 1 else 1.0)
    raw.append((w, torch.randn(dims[i + 1], generator=g) * 0.1, torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1))
pack = ops.pack_decode_split(raw).to(dev)
for M in (1000, 2 ** 20, 2 ** 21 + 77, 16 * 2 ** 20):
    x = torch.relu(torch.randn(M, 32, generator=g)) * 2.0
    if REST:
        x.zero_()
    xin = ops.new_rows(M, 32, dev); xin.copy_(x.to(dev))
    out = ops.implicit_decode_split(xin, pack); torch.cuda.synchronize()
    reps = 20 if M < 2 ** 22 else 5
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [ops.implicit_decode_split(xin, pack, out=out) for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * (32 * NH + NH * NH + NH) * M
    print(f'M={M}: {ms:.4f} ms {fl / ms / 1e9:.1f} TF(eq) digest {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}', flush=True)
