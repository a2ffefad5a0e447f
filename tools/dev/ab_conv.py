import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev='cuda'
def run(B,G,C0,Cout,reps=3):
    x = torch.randn(B,G,G,G,C0, device=dev)
    a = torch.ones(B,C0,device=dev); d = torch.zeros(B,C0,device=dev)
    wp = torch.randn(27,C0//16,Cout,16,device=dev)*0.01
    for nt4 in (0,0):
        ops.conv3d_gcr(x,None,a,d,wp,Cout); torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.conv3d_gcr(x,None,a,d,wp,Cout)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/reps
        print(f'B={B} G={G} {C0}->{Cout} nt4={nt4}: {ms:.2f} ms  {54.0*C0*Cout*B*G**3/ms/1e9:.1f} TF')
run(4,128,128,128)
run(4,128,128,32)
run(4,128,32,32)
run(16,32,128,128)
run(16,32,384,128)
