"""dev: does the iso tail of batch k really run beside the dense path of batch k+1?  HIP events: end of dense(k) -> end of iso(k) lanes"""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, synthetic as S
from garmentnets_amd.batch import Batch
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
from garmentnets_amd.predict import PredictJob, predict_batch
from garmentnets_amd.common import marching_cubes_util as mcu
dev = 'cuda:0'
hp = S.default_hparams(grid=128, reduce_method='mean')
m = ConvImplicitWNFPipeline(**hp); m.load_state_dict(S.synthetic_state_dict(hp, 0)); m = m.to(dev).eval().requires_grad_(False)
ops.SPARSE_FIRST_CONV = False
x, pos, batch = S.synthetic_cloud(16, 6000, seed=0)
data = Batch(sizes=[6000] * 16, x=x, pos=pos, batch=batch).to(dev)
for _ in range(2): predict_batch(m, data, 128, 0.5)
def run(n, pipelined):
    evs = []; host = []; hfin = []
    prev = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        h0 = time.perf_counter(); job = PredictJob(m, data, 128, 0.5, bank=1 + (k & 1)); h1 = time.perf_counter(); host.append(round((h1 - h0) * 1e3, 1))
        lanes_done = []
        for st in job.state['job'].lanes_used:
            e = torch.cuda.Event(enable_timing=True); e.record(st); lanes_done.append(e)
        r = torch.cuda.Event(enable_timing=True); r.record(torch.cuda.current_stream())
        evs.append((r, lanes_done))
        if pipelined:
            if prev is not None:
                h0 = time.perf_counter(); prev.finish(); hfin.append(round((time.perf_counter() - h0) * 1e3, 1))
            prev = job
        else:
            job.finish()
    if prev is not None: prev.finish()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    gaps = [max(r.elapsed_time(e) for e in ld) for r, ld in evs if ld]
    print('host ms in PredictJob():', host, 'in finish():', hfin)
    print('pipelined' if pipelined else 'sequential', f'{dt*1e3:.1f} ms/step; dense end -> last iso lane end (ms):', [round(g, 1) for g in gaps])
run(6, True); run(6, True); run(6, False); run(6, True)
