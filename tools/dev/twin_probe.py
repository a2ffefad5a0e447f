"""dev-only: why does the surface decoder's gated fp32 twin run on the benchmark's own checkpoint?  Prints, per decoder of the bench model, the
pack's smax, its largest scaled biases, and per garment the natural scale s0 (largest channel rms of the pre-final volume -> [1, 2)), the clamped
scale and the `unsafe` flag gn_decoder_input_scale derives (csrc/decode_split.hip)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from garmentnets_amd.arith import Arith
from garmentnets_amd.batch import Batch
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline

kind = sys.argv[1] if len(sys.argv) > 1 else "planted"
hp, sd, shard, _ = bench.bench_inputs(16, 6000, 128, "mean", kind)
dev = torch.device("cuda", 0)
model = ConvImplicitWNFPipeline(**hp)
model.load_state_dict(sd)
model = model.to(dev).eval().requires_grad_(False)
model.arith = Arith.named("f16x2", "f16x2", sparse_first_conv=False)
data = Batch(sizes=shard.sizes, x=shard.x, pos=shard.pos, batch=shard.batch).to(dev)
with torch.no_grad():
    u3 = model.unet3d_forward(model.pointnet2_forward(data))
st = u3.pre_stats
rms = (st[1] / st[2]).sqrt()                              # [B][32]
print("pre-final volume: per-garment largest channel rms", [f"{v:.3e}" for v in rms.amax(dim=1).tolist()])
print("                  per-garment smallest channel rms", [f"{v:.3e}" for v in rms.amin(dim=1).tolist()])
for name in ("volume_decoder", "surface_decoder"):
    dec = getattr(model, name)
    layers = dec.folded_pack(u3.final_conv)
    if layers is None:
        print(name, "not foldable"); continue
    pk = layers[3]
    tab = pk.tab.cpu()
    t1 = tab[:256]
    print(f"{name}: out={pk.out_channels} smax=2^{math.log2(pk.smax):.0f} max|b1 scaled|={float(t1.abs().max()):.3e}")
    xs = u3.input_scales(pk.smax).cpu()
    for b in range(xs.shape[0]):
        s0 = 2.0 ** (1 - math.frexp(float(rms[b].amax()))[1])
        print(f"   garment {b:2d}: s0=2^{math.log2(s0):.0f} s=2^{math.log2(float(xs[b, 0])):.0f} unsafe={int(xs[b, 2])}")
