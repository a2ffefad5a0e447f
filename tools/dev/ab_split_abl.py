"""dev-only: time ablated builds of the split conv (SP_ABL bit flags) to find what bounds it.
build: for k in 0 1 2 3 4 7; do hipcc ... -DSP_ABL=$k unet_split.hip points.hip -shared -o tools/dev/_build/libsplit_$k.so; done"""
import sys, os, ctypes, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from garmentnets_amd import ops
dev = 'cuda'
P_ = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
def run(lib, tag, B, G, C0, Cout, P, reps=3):
    x = torch.randn(B, G, G, G, C0, device=dev)
    a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
    w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
    wps = ops.pack_conv_weight_split(w, P).to(dev)
    out = torch.empty(B, G, G, G, Cout, device=dev)
    fl = 54.0 * C0 * Cout * B * G ** 3
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.gn_conv3d_gcr_split(P_(x), C0, None, 0, P_(a), P_(d), P_(wps), P, B, G, G, G, Cout, 1, P_(out), None, None, st)
    assert f() == 0; torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{tag:8s} B={B} G={G} {C0}->{Cout} P={P}: {ms:7.2f} ms {fl/ms/1e9:6.1f} TF(eq)  {fl/ms/1e9*(6 if P==3 else 3)/1e3:5.2f} PF bf16', flush=True)
for k in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(root, 'tools/dev/_build', f'libsplit_{k}.so'))
    for P in (3, 2):
        run(lib, f'abl={k}', 4, 128, 128, 128, P)
        run(lib, f'abl={k}', 4, 128, 128, 32, P)
