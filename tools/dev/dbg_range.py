import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
DEV='cuda'
g = torch.Generator().manual_seed(9)
B, C0, Cout, (D, H, W) = 1, 32, 32, (4, 8, 8)
x0 = torch.randn(B, C0, D, H, W, generator=g)
w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
st = ops.channel_stats(s0)
for gain in (1e4, 1e5, 1e6):
    gamma, beta = torch.full((C0,), gain), torch.zeros(C0)
    a, d = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    ref64 = F.relu(F.conv3d(F.group_norm(x0.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    out = ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), Cout).permute(0, 4, 1, 2, 3).cpu().double()
    print(gain, 'finite', bool(torch.isfinite(out).all()), 'nan', int(torch.isnan(out).sum()), 'inf', int(torch.isinf(out).sum()), 'ref max', ref64.max().item(), 'out max', out[torch.isfinite(out)].max().item() if torch.isfinite(out).any() else None,
          'relerr', ((out-ref64).abs().max()/ref64.abs().max()).item())
