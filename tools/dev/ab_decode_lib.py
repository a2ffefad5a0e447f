"""dev-only: time gn_implicit_decode_split of two library builds (K0 = 128 and 32)"""
import sys, os, ctypes, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from garmentnets_amd import ops
dev = 'cuda'; M = 1 << 18
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
for K0 in (128, 32):
    g = torch.Generator().manual_seed(0)
    dims = [K0, 256, 256, 1]
    raw = [(torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5, torch.randn(dims[i + 1], generator=g) * 0.1,
            torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1) for i in range(3)]
    xin = ops.new_rows(M, K0, dev); xin.copy_(torch.randn(M, K0, generator=g).to(dev))
    pk = ops.pack_decode_split(raw).to(dev)
    for path in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.join(root, path))
        out = torch.empty(M, 1, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: lib.gn_implicit_decode_split(P_(xin), K0, ctypes.c_int64(M), P_(pk.wpack), P_(pk.tab), ctypes.c_float(pk.inv1), ctypes.c_float(pk.inv2),
                                                 K0, 256, 256, 1, P_(out), 1, st)
        assert f() == 0; torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
        print(f'K0={K0} {path}: {e0.elapsed_time(e1)/10:.4f} ms', flush=True)
