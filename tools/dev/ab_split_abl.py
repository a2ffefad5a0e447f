"""dev-only: time ablated builds of the split conv (SP_ABL bit flags: 1 no B DMA, 2 no halo staging after slice 0, 4 no A/B
fragment re-reads, 8 no per-tap barrier) to find what bounds it.  Builds: tools/dev/_build/libsplit_<k>.so"""
import sys, os, ctypes, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from garmentnets_amd import ops
dev = 'cuda'
P_ = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
def run(lib, tag, B, G, C0, Cout, mode, reps=3):
    x = torch.randn(B, G, G, G, C0, device=dev)
    a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
    w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
    pk = ops.pack_conv_weight_split(w, mode).to(dev)
    out = torch.empty(B, G, G, G, Cout, device=dev)
    fl = 54.0 * C0 * Cout * B * G ** 3
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.gn_conv3d_gcr_split(P_(x), C0, None, 0, P_(a), P_(d), P_(pk.tensor), mode, P_(pk.out_scale), None, B, G, G, G, Cout, 1, P_(out), None, None, st)
    assert f() == 0; torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{tag:8s} B={B} G={G} {C0}->{Cout} mode={mode}: {ms:7.2f} ms {fl/ms/1e9:6.1f} TF(eq)  {fl/ms/1e9*(6 if mode==3 else 3)/1e3:5.2f} PF', flush=True)
for k in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(root, 'tools/dev/_build', f'libsplit_{k}.so'))
    run(lib, f'abl={k}', 4, 128, 128, 128, 4)
    run(lib, f'abl={k}', 4, 128, 128, 32, 4)
