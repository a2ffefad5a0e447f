"""dev: where the time after the WNF lattice goes (iso enqueue on the device, host sync, tail), HIP events + host clocks"""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, synthetic as S, predict as PR
from garmentnets_amd.batch import Batch
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
from garmentnets_amd.common import marching_cubes_util as mcu
dev = 'cuda:0'
hp = S.default_hparams(grid=128, reduce_method='mean')
m = ConvImplicitWNFPipeline(**hp); m.load_state_dict(S.synthetic_state_dict(hp, 0)); m = m.to(dev).eval().requires_grad_(False)
ops.SPARSE_FIRST_CONV = False
x, pos, batch = S.synthetic_cloud(16, 6000, seed=0)
data = Batch(sizes=[6000] * 16, x=x, pos=pos, batch=batch).to(dev)
for _ in range(3): PR.predict_batch(m, data, 128, 0.5)
E = lambda: torch.cuda.Event(enable_timing=True)
for rep in range(3):
    with torch.no_grad():
        torch.cuda.synchronize()
        p2 = m.pointnet2_forward(data); u3 = m.unet3d_forward(p2)
        wnf = m.volume_lattice_forward(u3, 128)["pred_volume"]
        e0 = E(); e0.record()
        job = mcu.IsoBatchJob(128, 0.5, 0.5, "ascent"); job.enqueue(wnf)
        e1 = E(); e1.record()
        h0 = time.perf_counter()
        meshes = job.finish()
        h1 = time.perf_counter()
        e2 = E(); e2.record()
        q = job.padded_queries()
        warp = m.surface_decoder_forward(u3, q)["out_features"]
        e3 = E(); e3.record()
        torch.cuda.synchronize()
        h2 = time.perf_counter()
    print(f'iso device {e0.elapsed_time(e1):.2f} ms | finish (sync + slices) {e1.elapsed_time(e2):.2f} ms, host {1e3*(h1-h0):.2f} | surface decode {e2.elapsed_time(e3):.2f} ms, host total {1e3*(h2-h0):.2f}; q {tuple(q.shape)}')
