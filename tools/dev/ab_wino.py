"""dev-only: the Winograd F(2,3)-along-x form of the 128-wide conv (csrc/unet_wino.hip) against the direct form (conv3d_split_wide_kernel):
error against fp64 on small volumes (both operand forms), then time / TFLOP/s-eq on the first encoder layer's shape (128 -> 128 at 128^3) with
zero, scattered (affine-in-weights) and N(0,1) operands.  usage: ab_wino.py [check|time|all]"""
import hashlib, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
dev = 'cuda'
what = sys.argv[1] if len(sys.argv) > 1 else 'all'


def check(B, dims, C0, Cout, seed, scattered):
    g = torch.Generator().manual_seed(seed)
    D, H, W = dims
    x = torch.randn(B, C0, D, H, W, generator=g)
    if scattered:
        x = x * (torch.rand(B, 1, D, H, W, generator=g) < 0.05)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    gamma, beta = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    ref = F.relu(F.conv3d(F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    s0 = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    st = ops.channel_stats(s0)
    a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(dev), beta.to(dev), with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(dev), beta.to(dev))
    cl = lambda t: t.permute(0, 4, 1, 2, 3).cpu().double()
    e = {}
    e['fp32-mfma'] = (cl(ops.conv3d_gcr(s0, None, a0, d0, ops.pack_conv_weight(w).to(dev), Cout)) - ref).abs().max().item()
    e['direct literal'] = (cl(ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, 4).to(dev), Cout, act_inv=inv)) - ref).abs().max().item()
    yw, (sm, sq, V) = ops.conv3d_gcr_split_wino(s0, a, d, ops.pack_conv_weight_split_wino(w).to(dev), Cout, act_inv=inv, with_stats=True)
    e['wino literal'] = (cl(yw) - ref).abs().max().item()
    stat_err = (sm.cpu() - cl(yw).sum(dim=(2, 3, 4))).abs().max().item()
    wd = w.to(dev).contiguous()
    e['direct at-rest'] = (cl(ops.conv3d_gcr_split_persample(s0, ops.conv_affine_pack(wd, a0, d0, st))) - ref).abs().max().item()
    e['wino at-rest'] = (cl(ops.conv3d_gcr_split_persample(s0, ops.conv_affine_pack(wd, a0, d0, st, wino=True))) - ref).abs().max().item()
    print(f'B={B} {dims} {C0}->{Cout} {"scattered" if scattered else "dense"}: max|ref| {ref.abs().max().item():.2f}  ' + '  '.join(f'{k} {v:.2e}' for k, v in e.items())
          + f'  stats err {stat_err:.1e}', flush=True)


def bench(B, G, C0, Cout, kind, reps=6, only=None):
    g = torch.Generator().manual_seed(C0 + Cout + G)
    if kind == 'zeros':
        x = torch.zeros(B, G, G, G, C0, device=dev)
    elif kind == 'randn':
        x = torch.randn(B, G, G, G, C0, generator=g).to(dev)
    else:                                           # scattered: 0.25 % of the cells occupied
        x = (torch.randn(B, G, G, G, C0, generator=g) * (torch.rand(B, G, G, G, 1, generator=g) < 0.0025)).to(dev)
    w = (torch.randn(Cout, C0, 3, 3, 3, generator=g) * 0.05)
    gamma, beta = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g) * 0.1
    st = ops.channel_stats(x)
    a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(dev), beta.to(dev), with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(dev), beta.to(dev))
    if kind == 'zeros':
        d = torch.zeros_like(d); d0 = torch.zeros_like(d0)
    pk, pw = ops.pack_conv_weight_split(w, 4).to(dev), ops.pack_conv_weight_split_wino(w).to(dev)
    wd = w.to(dev).contiguous()
    prep, prepw = ops.conv_affine_pack(wd, a0, d0, st), ops.conv_affine_pack(wd, a0, d0, st, wino=True)
    runs = {'direct literal': lambda: ops.conv3d_gcr_split(x, None, a, d, pk, Cout, with_stats=True, act_inv=inv),
            'wino   literal': lambda: ops.conv3d_gcr_split_wino(x, a, d, pw, Cout, with_stats=True, act_inv=inv),
            'direct at-rest': lambda: ops.conv3d_gcr_split_persample(x, prep, with_stats=True),
            'wino   at-rest': lambda: ops.conv3d_gcr_split_persample(x, prepw, with_stats=True)}
    for name, f in runs.items():
        if only is not None and name.split()[1] != only:
            continue
        y, _ = f(); torch.cuda.synchronize()
        kern = _lib.load().gn_last_kernel().decode()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]
        print(f'B={B} {G}^3 {C0}->{Cout} {kind:9s} {name}: {ms:8.3f} ms {54.0 * C0 * Cout * B * G ** 3 / ms / 1e9:7.1f} TF(eq) digest {dig} [{kern}]', flush=True)
        del y
        time.sleep(0.5)


if what in ('check', 'all'):
    check(2, (8, 16, 16), 32, 128, 1, False)
    check(2, (8, 16, 16), 32, 128, 2, True)
    check(1, (4, 8, 8), 128, 128, 3, False)
    check(2, (12, 8, 24), 64, 256, 4, False)
    check(1, (16, 16, 16), 128, 128, 5, True)
if what in ('time', 'all'):
    for rep in range(2):
        bench(8, 128, 128, 128, 'zeros')
        bench(8, 128, 128, 128, 'scattered')
        bench(8, 128, 128, 128, 'randn')
    bench(16, 32, 128, 128, 'randn', reps=10)
if what == 'prof_scattered':
    bench(8, 128, 128, 128, 'scattered', reps=3, only='at-rest')
if what == 'prof_randn':
    bench(8, 128, 128, 128, 'randn', reps=3, only='literal')
if what == 'abl':                     # timing only (ablation builds give wrong results): the at-rest form on scattered and N(0,1) operands
    bench(8, 128, 128, 128, 'scattered', reps=6, only='at-rest')
    bench(8, 128, 128, 128, 'randn', reps=6, only='at-rest')
