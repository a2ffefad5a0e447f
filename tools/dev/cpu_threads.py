import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pipeline as P
from garmentnets_amd import synthetic as S
hp = S.default_hparams(grid=64); sd = S.synthetic_state_dict(hp, 0)
x = torch.randn(1,128,64,64,64)
print('cpu_count', os.cpu_count())
for nt in (16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    with torch.no_grad():
        P.unet3d(sd, hp['unet3d_params'], x[:, :, :16, :16, :16])
        t=time.time(); P.unet3d(sd, hp['unet3d_params'], x); dt=time.time()-t
        vol = torch.randn(1,128,32,32,32)
        t=time.time(); P.decode_volume(sd, vol, 64); dt2=time.time()-t
    print(nt, 'unet64 %.2fs decode64^3 %.2fs'%(dt, dt2), flush=True)
