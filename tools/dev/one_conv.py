"""dev-only: a few launches of one split conv shape (for rocprofv3 --pmc runs): one_conv.py C0 Cout [B] [G]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
C0, Cout = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
G = int(sys.argv[4]) if len(sys.argv) > 4 else 128
x = torch.randn(B, G, G, G, C0, device='cuda'); a = torch.ones(B, C0, device='cuda'); d = torch.zeros(B, C0, device='cuda')
w = torch.randn(Cout, C0, 3, 3, 3) * 0.02
pk = ops.pack_conv_weight_split(w, 4).to('cuda')
for _ in range(3): ops.conv3d_gcr_split(x, None, a, d, pk, Cout)
torch.cuda.synchronize()
