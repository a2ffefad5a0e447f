import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, synthetic as S
from garmentnets_amd.networks.conv_implicit_wnf import ImplicitWNFDecoder
dev='cuda'
dec = ImplicitWNFDecoder((128,256,256,1)).to(dev).eval()
layers = dec.packed()
G=128; Q=128
vol = torch.randn(G,G,G,128, device=dev)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
M=Q**3
out = torch.empty(M,1,device=dev)
print('fused lattice        %.2f ms'%t(lambda: ops.implicit_decode(vol, layers, Q=Q, m0=0, M=M, out=out)))
X = ops.new_rows(M,128,dev)
print('sample only          %.2f ms'%t(lambda: ops.trilinear_sample(vol, Q=Q, m0=0, M=M, out=X)))
print('mlp only (presampled)%.2f ms  -> %.1f TF'%((lambda ms:(ms, 417.4/ms))(t(lambda: ops.implicit_decode(None, layers, M=M, out=out, xin=X)))))
q = torch.rand(M,3,device=dev)
print('fused random queries %.2f ms'%t(lambda: ops.implicit_decode(vol, layers, query=q, out=out)))
vol32 = torch.randn(32,32,32,128, device=dev)
print('fused lattice G=32   %.2f ms'%t(lambda: ops.implicit_decode(vol32, layers, Q=Q, m0=0, M=M, out=out)))
print('sample only G=32     %.2f ms'%t(lambda: ops.trilinear_sample(vol32, Q=Q, m0=0, M=M, out=X)))
