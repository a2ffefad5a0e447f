"""dev-only: practical fill / copy bandwidth for a 134 MB buffer (the decoder's chunk buffer)"""
import torch
a = torch.empty(262144, 128, device='cuda'); b = torch.randn(262144, 128, device='cuda')
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print(f"fill 134 MB: {ms:.4f} ms  {134.2/ms:.0f} GB/s write")
ms = t(lambda: a.copy_(b)); print(f"copy 134 MB: {ms:.4f} ms  {134.2/ms:.0f} GB/s write + same read")
big = torch.empty(1 << 30, device='cuda', dtype=torch.uint8)
ms = t(lambda: big.fill_(1)); print(f"fill 1 GB: {ms:.4f} ms  {1073.7/ms:.0f} GB/s")
