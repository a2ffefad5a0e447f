"""dev-only: the non-default shapes the limits table of DESIGN.md 8 lists, timed -- the decoder with the reference's CLASS-default widths
(networks/conv_implicit_wnf.py:122: nn_channels=(128, 512, 512, 1); the shipped config uses (128, 256, 256, 1)) goes through the fused fp32-MFMA decoder
kernel (gn_implicit_decode), not the split-operand one; an edge MLP gn_sa_fused is not instantiated for goes through the unfused chain."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import arith as AR, ops, synthetic as S
from garmentnets_amd.networks.conv_implicit_wnf import ImplicitWNFDecoder
from garmentnets_amd.components.pointnet2 import SAModule, Segments
from garmentnets_amd.components.mlp import MLP
dev = 'cuda'
torch.manual_seed(0)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
vol = torch.randn(1, 128, 64, 64, 64, device=dev).relu()
q = torch.rand(1, 2 ** 21, 3, device=dev)
for ch in ((128, 256, 256, 1), (128, 512, 512, 1)):
    dec = ImplicitWNFDecoder(nn_channels=ch).to(dev).eval().requires_grad_(False)
    for name, ar in (("f16x2", AR.Arith.named("f16x2", "f16x2")), ("fp32", AR.Arith.named("fp32", "fp32"))):
        with torch.no_grad():
            ms = timed(lambda: dec(vol, q, arith=ar))
        fl = 2.0 * (ch[0] * ch[1] + ch[1] * ch[2] + ch[2] * ch[3]) * q.shape[1]
        print(f"decoder {ch} {name}: {ms:.3f} ms per 2^21 queries (sampler included), {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
# the same two decoders the way the pipeline runs them: on the UNet's 32-channel PRE-final volume with the final 1x1x1 convolution folded into the first layer
# (UNetResult / folded_pack) -- both hidden widths then have a split-operand pack (csrc/decode_split.hip: the 256-wide kernel, implicit_decode_split512_kernel)
from garmentnets_amd.networks.conv_implicit_wnf import UNetResult
from garmentnets_amd.components.unet3d import FinalConv1x1
fc = FinalConv1x1(32, 128, 1).to(dev).eval().requires_grad_(False)
pre = torch.randn(1, 64, 64, 64, 32, device=dev).relu()
res = {}
for ch in ((128, 256, 256, 1), (128, 512, 512, 1)):
    dec = ImplicitWNFDecoder(nn_channels=ch).to(dev).eval().requires_grad_(False)
    for name, ar in (("f16x2", AR.Arith.named("f16x2", "f16x2")), ("fp32", AR.Arith.named("fp32", "fp32"))):
        with torch.no_grad():
            ms = timed(lambda: dec.run_on(UNetResult(pre, fc), q, arith=ar))
        fl = 2.0 * (32 * ch[1] + ch[1] * ch[2] + ch[2] * ch[3]) * q.shape[1]
        res[(ch[1], name)] = (ms, fl)
        print(f"folded decoder [32,{ch[1]},{ch[2]},{ch[3]}] {name}: {ms:.3f} ms per 2^21 queries (sampler included), {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
(m2, f2), (m5, f5) = res[(256, "f16x2")], res[(512, "f16x2")]
print(f"f16x2: time ratio 512 / 256 = {m5 / m2:.2f}, FLOP ratio {f5 / f2:.2f} -> {m5 / m2 / (f5 / f2):.2f}x its FLOP ratio", flush=True)
# set abstraction with a non-shipped edge MLP [3+3, 32, 64, 128] (unfused chain) next to the shipped one (fused kernel)
x, pos, batch = S.synthetic_cloud(16, 6000, seed=1)
x, pos = x.to(dev), pos.to(dev)
seg = Segments([6000] * 16, dev)
for dims in ([6, 64, 64, 128], [6, 32, 64, 128], [6, 48, 64, 128]):      # shipped (fused) / instantiated since round 5 (fused) / not instantiated (unfused chain)
    sa = SAModule(0.5, 0.05, MLP(dims, batch_norm=True)).to(dev).eval().requires_grad_(False)
    with torch.no_grad():
        ms = timed(lambda: sa(x, pos, seg))
    print(f"SAModule edge MLP {dims}: {ms:.3f} ms per 16 x 6000 points (fps + ball query + PointConv)", flush=True)
