"""dev-only: per-layer table of the UNet's convolutions in the benchmark step (shape, kernel variant, ms, TFLOP/s-eq)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib, synthetic as S
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline

dev = torch.device("cuda:0")
hp = S.default_hparams(grid=128, reduce_method="mean")
sd = S.synthetic_state_dict(hp, 0, planted_nocs=True)
model = ConvImplicitWNFPipeline(**hp); model.load_state_dict(sd); model = model.to(dev).eval().requires_grad_(False)
x, pos, batch = S.synthetic_cloud(16, 6000, 0, colour="position")
from garmentnets_amd.batch import Batch
data = Batch(sizes=[6000] * 16, x=x, pos=pos, batch=batch).to(dev)
model.arith = model.arith.replace(sparse_first_conv=(os.environ.get("PER_LAYER_SPARSE", "0") == "1"))
rec = []
def wrap(name, fn, desc):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record()
        rec.append((desc(out, *a, **k), e0, e1))
        return out
    return f
def d_split(out, src0, src1, a, d, pk, cout, *r, **k):
    B, D, H, W, C0 = src0.shape; C1 = 0 if src1 is None else src1.shape[-1]
    return (f"split {C0}+{C1}->{cout} @{D} {'partial' if k.get('partial') is not None else ''}", _lib.load().gn_last_kernel().decode(), 54.0 * (C0 + C1) * cout * B * D * H * W)
def d_ps(out, src, prep, *r, **k):
    B, D, H, W, C = src.shape
    return (f"at-rest {C}->{prep.cout} @{D} {'partial' if k.get('partial') is not None else ''}", _lib.load().gn_last_kernel().decode(), 54.0 * C * prep.cout * B * D * H * W)
def d_up(out, src1, a1, d1, pack, cout, **k):
    B, Dc, Hc, Wc, C1 = src1.shape
    return (f"upconv {C1}->8x{cout} @{Dc}", "upconv_partial_kernel", 2.0 * 64 * C1 * cout * B * Dc * Hc * Wc)
ops.conv3d_gcr_split = wrap("s", ops.conv3d_gcr_split, d_split)
ops.conv3d_gcr_split_persample = wrap("p", ops.conv3d_gcr_split_persample, d_ps)
ops.upconv_partial = wrap("u", ops.upconv_partial, d_up)
with torch.no_grad():
    for it in range(3):
        rec.clear()
        p2 = model.pointnet2_forward(data)
        u3 = model.unet3d_forward(p2)
        torch.cuda.synchronize()
tot = 0.0
for (name, kern, fl), e0, e1 in rec:
    ms = e0.elapsed_time(e1); tot += ms
    print(f"{name:34s} {kern:38s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF-eq")
print(f"sum {tot:.2f} ms")
