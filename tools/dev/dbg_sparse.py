import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from garmentnets_amd import ops, synthetic as S
from garmentnets_amd.components.unet3d import SingleConv
DEV = "cuda"
for G, mode in ((32, 4), (32, 2), (20, 4)):
    g = torch.Generator().manual_seed(G)
    B, C, Cout = 3, 128, 128
    cells = [torch.tensor([[0, 0, 0], [G - 1, G - 1, G - 1], [0, G - 1, 5], [G // 2, G // 2, G // 2], [3, 8, 9]]), torch.zeros((0, 3), dtype=torch.int64), torch.randint(0, G, (200, 3), generator=g)]
    flat = torch.cat([(((b * G + c[:, 0]) * G + c[:, 1]) * G + c[:, 2]) for b, c in enumerate(cells)]).to(torch.int32)
    feats = torch.randn(flat.numel(), C, generator=g)
    vol, stats = ops.grid_scatter(feats.to(DEV), flat.to(DEV), B, (G, G, G), "max", with_stats=True)
    conv = SingleConv(C, Cout)
    conv.load_state_dict({k: S.synthetic_tensor("c." + k, tuple(v.shape), 1) for k, v in conv.state_dict().items()})
    conv = conv.to(DEV)
    ops.CONV_MODE, ops.SPARSE_FIRST_CONV = mode, True
    y_s, (s_s, q_s, V) = conv.run(vol, None, stats, None, sparse_flat=flat.to(DEV))
    flags = ops.grid_tile_flags(flat.to(DEV), B, (G, G, G))
    y_d, (s_d, q_d, _) = conv.run(vol, None, stats, None)
    # dense through the wide kernel: all tiles active
    ops.SPARSE_FIRST_CONV = True
    diff = (y_s - y_d).abs()
    print(G, mode, "y equal", torch.equal(y_s, y_d), "max diff", float(diff.max()), "stats equal", torch.equal(s_s, s_d), torch.equal(q_s, q_d))
    if not torch.equal(y_s, y_d):
        idx = (diff > 0).nonzero()
        print("  n diff", idx.shape[0], "first", idx[:5].tolist(), "values", [ (float(y_s[tuple(i)]), float(y_d[tuple(i)])) for i in idx[:5]])
        tz = (G + 3) // 4; ty = (G + 7) // 8; tx = ty
        for i in idx[:5]:
            b, z, y, x, c = i.tolist()
            t = ((y // 8) * tx + (x // 8)) * tz + z // 4
            print("   tile active?", int(flags[b, t]))
    print("  stats diff", float((s_s - s_d).abs().max()), float((q_s - q_d).abs().max()))
