"""dev-only: brick sampler ablations (TB_ABL bits: 1 no DMA staging, 2 no corner reads/FMAs, 4 no stores); libs tools/dev/_build/libtb_<k>.so"""
import sys, os, ctypes, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = Q = 128
vol = torch.randn(G, G, G, 128, device='cuda')
M = 16 * Q * Q
out = torch.empty(M, 128, device='cuda')
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
for k in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(root, 'tools/dev/_build', f'libtb_{k}.so'))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.gn_trilinear_sample(P_(vol), G, G, G, 128, None, Q, ctypes.c_int64(16 * Q * Q), ctypes.c_int64(M), P_(out), 128, st)
    assert f() == 0; torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
    print(f'abl={k}: {e0.elapsed_time(e1)/10:.4f} ms', flush=True)
