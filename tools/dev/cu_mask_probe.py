"""dev-only: the x-strip conv (128 -> 32 at 128^3, B = 8; many short workgroups, no persistent grid) on a stream whose CU mask leaves N of the 256 CUs
enabled (hipExtStreamCreateWithCUMask): if the matrix kernels sit at the socket's power limit, fewer CUs at a higher clock lose less than their share"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
hip = ctypes.CDLL('libamdhip64.so')
g = torch.Generator().manual_seed(3)
B, G, C0, Cout = 8, 128, 128, 32
kinds = {'randn': torch.randn(B, G, G, G, C0, generator=g).to(dev), 'zeros': torch.zeros(B, G, G, G, C0, device=dev)}
a = (torch.rand(B, C0, generator=g) + 0.5).to(dev); d = (torch.randn(B, C0, generator=g) * 0.1).to(dev)
inv = torch.full((B,), 0.5, device=dev)
pk = ops.pack_conv_weight_split(torch.randn(Cout, C0, 3, 3, 3, generator=g) * 0.05, 4).to(dev)
torch.cuda.synchronize()


def masked_stream(words):
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


FULL = 0xFFFFFFFF
masks = {'256 CUs': [FULL] * 8,
         '224 CUs (one word of 32 off)': [FULL] * 7 + [0],
         '224 CUs (4 bits off per word)': [0x0FFFFFFF] * 8,
         '192 CUs (8 bits off per word)': [0x00FFFFFF] * 8,
         '128 CUs (16 bits off per word)': [0x0000FFFF] * 8}
for name, words in masks.items():
    st = masked_stream(words)
    for kind, x in kinds.items():
        dz = torch.zeros_like(d) if kind == 'zeros' else d
        with torch.cuda.stream(st):
            f = lambda: ops.conv3d_gcr_split(x, None, a, dz, pk, Cout, relu=True, with_stats=True, act_inv=inv)
            f(); st.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st); [f() for _ in range(5)]; e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f'{name:34s} {kind}: {ms:.3f} ms  {54.0 * C0 * Cout * B * G ** 3 / ms / 1e9:.1f} TF(eq)', flush=True)
