"""Calibration of rocprofv3 FETCH_SIZE for the conv halo access pattern (64-byte pieces at a channel-count stride):
one 4x8x8 tile per sample (all halo voxels outside the volume are zero-filled without a read), so the kernel reads
every input element exactly once: known bytes = B*256*Cin*4 (+ weights)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
for C0, Cout, B in ((128, 64, 4096), (32, 32, 8192)):
    x = torch.randn(B, 4, 8, 8, C0, device=dev)
    a = torch.ones(B, C0, device=dev); d = torch.zeros(B, C0, device=dev)
    wp = ops.pack_conv_weight(torch.randn(Cout, C0, 3, 3, 3) * 0.02).to(dev)
    ops.conv3d_gcr(x, None, a, d, wp, Cout); torch.cuda.synchronize()
    print(f'CALIB C0={C0} Cout={Cout} B={B} known_input_bytes={x.numel()*4} out_bytes={B*256*Cout*4}')
