"""dev-only: a pure matrix-core burn at a chosen duty cycle, for the power / clock comparison of profiles/r03_power*: launches the 128 -> 128
split conv (the dominant kernel) on a zero volume and on an N(0,1) volume back to back for a few seconds each, so that the power trace shows
what the SAME instruction stream draws with and without operand toggling."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, _lib
B, G, C = 4, 128, 128
COUT = int(sys.argv[1]) if len(sys.argv) > 1 else C          # 128: the 128-wide kernel; 32: the x-strip kernel
w = torch.randn(COUT, C, 3, 3, 3) * 0.02
pk = ops.pack_conv_weight_split(w, 4).to('cuda')
a = torch.ones(B, C, device='cuda'); d = torch.zeros(B, C, device='cuda')
for name, x in (("zeros", torch.zeros(B, G, G, G, C, device='cuda')), ("randn", torch.randn(B, G, G, G, C, device='cuda'))):
    ops.conv3d_gcr_split(x, None, a, d, pk, COUT); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 6.0:
        for _ in range(10): ops.conv3d_gcr_split(x, None, a, d, pk, COUT)
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: t=[{t0:.2f}, {time.time():.2f}] {n} launches, {ms:.3f} ms each, {54.0*C*COUT*B*G**3/ms/1e9:.1f} TFLOP/s-eq [{_lib.load().gn_last_kernel().decode()}]", flush=True)
    time.sleep(2.0)
