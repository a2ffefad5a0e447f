"""dev-only A/B on one box: the last decoder's first convolution (32 + 64 -> 32 at 128^3, B = 16, polyphase) with the skip connection's
full-resolution launch in the literal form vs the affine-in-weights form (SingleConv.run rest0=)"""
import torch
from garmentnets_amd import ops, synthetic as S
from garmentnets_amd.components.unet3d import SingleConv

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, C0, C1, Cout, G = 16, 32, 64, 32, 128
rest = torch.rand(B, C0, generator=g)
x0 = (torch.zeros(B, G, G, G, C0) + rest[:, None, None, None, :])
n = 30000                                                   # ~ cells within 2 voxels of a 5000-cell garment surface
for b in range(B):
    idx = torch.randint(0, G, (n, 3), generator=g)
    x0[b, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.rand(n, C0, generator=g) * 2
x0 = x0.to(dev)
x1 = (torch.randn(B, G // 2, G // 2, G // 2, C1, generator=g) * 1.5).to(dev)
conv = SingleConv(C0 + C1, Cout)
conv.load_state_dict({k: S.synthetic_tensor("ab." + k, tuple(v.shape), 4) for k, v in conv.state_dict().items()})
conv = conv.to(dev)
rest = rest.to(dev)
st0, st1 = ops.channel_stats(x0), ops.channel_stats(x1)


def run(r0, n=12):
    for _ in range(3):
        conv.run(x0, x1, st0, st1, rest0=r0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        conv.run(x0, x1, st0, st1, rest0=r0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(2):
    print(f"literal {run(None):.3f} ms   skip at rest {run(rest):.3f} ms  (whole layer: partial + fine launch)")
