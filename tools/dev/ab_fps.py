import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, synthetic as S
from garmentnets_amd.components.pointnet2 import Segments
for B,n,ratio in ((1,6000,.5),(16,6000,.5),(16,3000,.25)):
    _,pos,_ = S.synthetic_cloud(B,n,3); pos=pos.cuda()
    seg=Segments([n]*B,'cuda'); cseg=Segments([ops.fps_count(n,ratio)]*B,'cuda')
    f=lambda: ops.fps(pos,seg.ptr,cseg.ptr,n,cseg.total)
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/3; print(f'fps B={B} n={n}: {ms:.3f} ms  {1e3*ms/(cseg.sizes[0]-1):.2f} us/step')
