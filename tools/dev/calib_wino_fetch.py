"""dev-only: calibration of rocprofv3 FETCH_SIZE for conv3d_split_wino_kernel (profiles/r01_fetch_calibration.txt's method): B samples of ONE 4 x 8 x 8 tile
each -- no halo reads, every input element is read exactly once: known bytes = B * 256 voxels * Cin * 4 (+ the weight pack, L2-resident after the first
workgroups).  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; tools/dev/calib_wino_fetch.sh prints reported KB, known KB and the factor."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops
dev = 'cuda'
B, C, Cout = 8192, 128, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 4, 8, 8, C, generator=g).to(dev)
w = torch.randn(Cout, C, 3, 3, 3, generator=g) * 0.05
pk = ops.pack_conv_weight_split_wino(w).to(dev)
a, d = torch.ones(B, C, device=dev), torch.zeros(B, C, device=dev)
for _ in range(3):
    ops.conv3d_gcr_split_wino(x, a, d, pk, Cout)
torch.cuda.synchronize()
print("known_input_KB", B * 256 * C * 4 / 1024, "pack_KB", pk.tensor.numel() * 2 / 1024)
