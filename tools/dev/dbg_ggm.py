import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
from oracle import pipeline as P
from garmentnets_amd import ops, synthetic as S
g = np.load('tests/golden/mc_golden.npz')
vol = g['smooth24_vol']
ref = O.ggm(vol, 0.5)
got = ops.ggm3d(torch.from_numpy(vol).cuda(), 0.5).cpu().numpy()
d = np.abs(got-ref)
print('ggm: nmismatch', int((got!=ref).sum()), 'of', got.size, 'max', d.max(), 'rel', (d/np.maximum(np.abs(ref),1e-30)).max())
i = np.unravel_index(np.argmax(d), d.shape); print(i, got[i], ref[i], np.float32(got[i]).view(np.uint32) - np.float32(ref[i]).view(np.uint32))
# unet fp64 check
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
for name in ['unet_g8','unet_g16']:
    gg = np.load(f'tests/golden/ref_{name}.npz'); G,B,seed = [int(v) for v in gg['meta']]
    hp = S.default_hparams(grid=G); sd = S.synthetic_state_dict(hp, seed)
    m = ConvImplicitWNFPipeline(**hp); m.load_state_dict(sd); m = m.cuda().eval()
    x = torch.randn(B,128,G,G,G, generator=torch.Generator().manual_seed(seed))
    y = m.unet_3d(x.cuda()).cpu().numpy()
    sd64 = {k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()}
    with torch.no_grad(): y64 = P.unet3d(sd64, hp['unet3d_params'], x.double()).numpy()
    print(name, 'gpu vs f64', np.abs(y-y64).max(), 'torch32 vs f64', np.abs(gg['y']-y64).max(), 'gpu vs torch32', np.abs(y-gg['y']).max(), 'scale', np.abs(y64).max(), 'rms', np.sqrt((y64**2).mean()))
