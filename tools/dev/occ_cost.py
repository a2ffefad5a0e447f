"""dev-only: cost of inactive / active tiles in the occupancy-aware launch of the first conv (128 -> 128 at 128^3, B = 16, scattered operand):
all tiles inactive, all active (= dense), the benchmark's own flags, and random flags of the same density"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
B, G, C = 16, 128, 128
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.zeros(B, G, G, G, C, device='cuda')
# a "garment": cells on a sphere shell -> clustered activity like the benchmark input
n = 5000
th = torch.rand(B, n, device='cuda', generator=g) * 6.283; ph = torch.acos(2 * torch.rand(B, n, device='cuda', generator=g) - 1)
r = 40.0
cz = (64 + r * torch.cos(ph)).long().clamp(0, G - 1); cy = (64 + r * torch.sin(ph) * torch.sin(th)).long().clamp(0, G - 1); cx = (64 + r * torch.sin(ph) * torch.cos(th)).long().clamp(0, G - 1)
flat = (((torch.arange(B, device='cuda')[:, None] * G + cz) * G + cy) * G + cx).reshape(-1)
x.view(-1, C)[flat] = torch.randn(flat.numel(), C, device='cuda', generator=g)
w = (torch.randn(C, C, 3, 3, 3) * 0.02).cuda()
st = ops.channel_stats(x)
a = torch.ones(B, C, device='cuda'); d = torch.full((B, C), 0.1, device='cuda')
prep = ops.conv_affine_pack(w, a, d, st, None)
flags_real = ops.grid_tile_flags(flat.to(torch.int32), B, (G, G, G), 1)
frac = float(flags_real.float().mean())
kconst = torch.zeros(B, 27, C, device='cuda')
def run(flags, name):
    kw = dict(tile_active=flags, kconst=kconst, kreach=1) if flags is not None else {}
    for _ in range(2): ops.conv3d_gcr_split_persample(x, prep, with_stats=True, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv3d_gcr_split_persample(x, prep, with_stats=True, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 5:8.3f} ms", flush=True)
run(None, "dense launch")
run(torch.ones_like(flags_real), "flags: all active")
run(torch.zeros_like(flags_real), "flags: all inactive")
run(flags_real, f"flags: shell garment ({frac:.3f} active)")
run((torch.rand(flags_real.shape, device='cuda', generator=g) < frac).to(torch.uint8), "flags: random, same density")
