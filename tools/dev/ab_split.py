import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev='cuda'
def run(B,G,C0,Cout,reps=3):
    x = torch.randn(B,G,G,G,C0, device=dev)
    a = torch.ones(B,C0,device=dev); d = torch.zeros(B,C0,device=dev)
    w = torch.randn(Cout,C0,3,3,3)*0.02
    wp = ops.pack_conv_weight(w).to(dev)
    fl = 54.0*C0*Cout*B*G**3
    def t(f):
        f(); torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(reps)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
    ms=t(lambda: ops.conv3d_gcr(x,None,a,d,wp,Cout)); print(f'B={B} G={G} {C0}->{Cout} fp32 : {ms:.2f} ms {fl/ms/1e9:.1f} TF(eq)')
    for P in (3,4,2):
        wps = ops.pack_conv_weight_split(w,P).to(dev)
        ms=t(lambda: ops.conv3d_gcr_split(x,None,a,d,wps,Cout)); print(f'   split mode={P}: {ms:.2f} ms {fl/ms/1e9:.1f} TF(eq)')
run(4,128,128,128)
run(4,128,128,32)
run(16,32,384,128)
