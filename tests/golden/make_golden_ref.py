#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own Python modules (build container only).

Run:  python tests/golden/make_golden_ref.py          (needs /root/reference; never runs on the GPU box)

The reference (/root/reference) imports only with sys.modules stubs for its absent third-party packages
(SURVEY.md Appendix D).  The five torch_geometric callables + torch_scatter.scatter are provided by stubs
that implement the *pinned definitions* of DESIGN.md section 3 (PyG/torch_cluster/torch_scatter are not
installable offline).  PointConv / global_max_pool / scatter are written here in edge-list / generic-torch form,
independently of the oracle's dense formulation and of the HIP kernels; fps, radius and knn_interpolate, however, ARE
the oracle's own C functions (oracle.fps / ball_query / knn_interpolate, lines 36-46 and 85-86 below): for those three
the goldens pin the COMPOSITION around them, not the operators -- tests/test_oracle_golden.py::
test_point_ops_against_plain_torch is what stands between a shared misreading of torch_cluster and a green suite.
Everything else -- MLP, PointNet2NOCS composition, heads,
NOCS arg-max post-processing, VirtualGrid index maths, VolumeFeatureAggregator, Abstract3DUNet,
ImplicitWNFDecoder, the 64^3-chunk decode loop -- is the reference's code executing on real torch.

Only DATA (inputs, expected outputs) is written to tests/golden/*.npz; no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle as O  # noqa: E402
from garmentnets_amd import synthetic as S  # noqa: E402

REF = "/root/reference"


# ------------------------------------------------------------------------------------------------ stubs
def _ptr(batch):
    return O.batch_to_ptr(batch.cpu().numpy())


def stub_fps(pos, batch=None, ratio=0.5, random_start=False):
    idx, _ = O.fps(pos.numpy(), _ptr(batch), ratio)
    return torch.from_numpy(idx)


def stub_radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1):
    n = x.shape[0]
    allp = torch.cat([x, y]).numpy()
    nbr, cnt = O.ball_query(allp, _ptr(batch_x), np.arange(n, n + y.shape[0]), _ptr(batch_y), r, max_num_neighbors)
    rows, cols = np.nonzero(nbr >= 0)
    return torch.from_numpy(rows.astype(np.int64)), torch.from_numpy(nbr[rows, cols].astype(np.int64))


class StubPointConv(torch.nn.Module):
    """PyG 1.7.2 PointConv semantics on an edge list (aggr='max', add_self_loops=True)."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops=True, **kw):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = global_nn
        self.add_self_loops = add_self_loops

    def forward(self, x, pos, edge_index):
        if isinstance(pos, torch.Tensor):
            pos = (pos, pos)
        xj_all = x[0] if isinstance(x, tuple) else x
        if self.add_self_loops:
            keep = edge_index[0] != edge_index[1]
            edge_index = edge_index[:, keep]
            m = pos[1].size(0)
            loops = torch.arange(m).unsqueeze(0).repeat(2, 1)
            edge_index = torch.cat([edge_index, loops], dim=1)
        j, i = edge_index[0], edge_index[1]
        msg = pos[0][j] - pos[1][i]
        if xj_all is not None:
            msg = torch.cat([xj_all[j], msg], dim=1)
        if self.local_nn is not None:
            msg = self.local_nn(msg)
        out = torch.zeros(pos[1].size(0), msg.size(1)).scatter_reduce(0, i.unsqueeze(1).expand_as(msg), msg, "amax", include_self=False)
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out


def stub_global_max_pool(x, batch, size=None):
    B = int(batch.max()) + 1 if size is None else size
    return torch.zeros(B, x.size(1)).scatter_reduce(0, batch.unsqueeze(1).expand_as(x), x, "amax", include_self=False)


def stub_knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, num_workers=1):
    return torch.from_numpy(O.knn_interpolate(x.numpy(), pos_x.numpy(), _ptr(batch_x), pos_y.numpy(), _ptr(batch_y), k))


def stub_scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert dim in (-1, src.dim() - 1)
    red = {"max": "amax", "mean": "mean", "sum": "sum", "add": "sum", "min": "amin"}[reduce]
    shape = list(src.shape)
    shape[-1] = dim_size
    return torch.zeros(shape, dtype=src.dtype).scatter_reduce(src.dim() - 1, index.expand_as(src), src, red, include_self=False)


class StubBatch:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_graphs(self):
        return int(self.batch.max()) + 1


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

    mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object)
    mod("wandb")
    mod("torch_scatter", scatter=stub_scatter)
    tg = mod("torch_geometric")
    tg.data = mod("torch_geometric.data", Batch=StubBatch, Data=StubBatch, DataLoader=object, Dataset=object)
    tg.datasets = mod("torch_geometric.datasets", ModelNet=object)
    tg.transforms = mod("torch_geometric.transforms")
    tg.nn = mod("torch_geometric.nn", PointConv=StubPointConv, fps=stub_fps, radius=stub_radius,
                global_max_pool=stub_global_max_pool, knn_interpolate=stub_knn_interpolate)
    mod("numba", jit=lambda *a, **k: (lambda f: f))
    sk = sys.modules.get("skimage") or mod("skimage")
    sk.transform = mod("skimage.transform", resize=None)
    sys.path.insert(0, REF)


def ref_pipeline(hp, sd):
    from networks.conv_implicit_wnf import ConvImplicitWNFPipeline
    kw = {k: hp[k] for k in ("pointnet2_params", "volume_agg_params", "unet3d_params", "volume_decoder_params",
                             "surface_decoder_params", "mc_surface_decoder_params", "mc_surface_loss_weight")}
    model = ConvImplicitWNFPipeline(**kw)
    ref_sd = model.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), (set(ref_sd) ^ set(sd))
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    model.requires_grad_(False)
    return model


def run_case(name, B, n_points, grid, reduce_method, Q, seed, store_full=True, stride=1):
    from components.gridding import VirtualGrid, ArraySlicer
    hp = S.default_hparams(grid=grid, reduce_method=reduce_method)
    sd = S.synthetic_state_dict(hp, seed=seed)
    model = ref_pipeline(hp, sd)
    x, pos, batch = S.synthetic_cloud(B, n_points, seed=seed)
    data = StubBatch(x=x, pos=pos, batch=batch)
    out = {}
    with torch.no_grad():
        p2 = model.pointnet2_forward(data)
        nd = p2["nocs_data"]
        u3 = model.unet3d_forward(p2)
        vol_in = model.volume_agg(nd)
        vol = u3["out_feature_volume"]
        # chunked volume decode exactly as predict.py:145-157, garment 0
        vg = VirtualGrid(grid_shape=(Q,) * 3)
        gp = vg.get_grid_points(include_batch=False)
        slicer = ArraySlicer(gp.shape, (64, 64, 64))
        wnf = torch.zeros(gp.shape[:-1])
        for i in range(len(slicer)):
            sl = tuple(slicer[i])
            q = gp[sl]
            r = model.volume_decoder_forward({"out_feature_volume": vol[0:1]}, q.reshape(1, -1, 3))
            wnf[sl] = r["pred_volume_value"].view(*q.shape[:-1])
        g = torch.Generator().manual_seed(seed + 77)
        sq = torch.rand(B, 257, 3, generator=g)
        sq[:, 0] = 0.0
        sq[:, 1] = 1.0
        sq[:, 2] = torch.tensor([0.0, 1.0, 0.5])
        surf = model.surface_decoder_forward(u3, sq)["out_features"]
        volq = model.volume_decoder_forward(u3, sq)
    sl = slice(None, None, stride)
    out.update(
        meta=np.array([B, n_points, grid, Q, seed, stride], np.int64), reduce_method=np.array(reduce_method),
        nocs_bin_idx=torch.argmax(p2["per_point_logits"].reshape(-1, 64, 3), dim=1).numpy().astype(np.int8),
        pred_nocs=nd.pos.numpy()[sl], pred_confidence=nd.pred_confidence.numpy()[sl],
        per_point_features=p2["per_point_features"].numpy()[sl], per_point_logits=p2["per_point_logits"].numpy()[sl],
        global_logits=p2["global_logits"].numpy(), global_feature=p2["global_feature"].numpy(),
        surf_query=sq.numpy(), surf_out=surf.numpy(), volq_out=volq["pred_volume_value"].numpy(),
        wnf_volume=wnf.numpy(),
        in_volume_sum=vol_in.double().sum(dim=(2, 3, 4)).numpy(), in_volume_abs=vol_in.double().abs().sum().numpy(),
        out_volume_sum=vol.double().sum(dim=(2, 3, 4)).numpy(), out_volume_abs=vol.double().abs().sum().numpy(),
    )
    if store_full:
        out.update(in_feature_volume=vol_in.numpy(), out_feature_volume=vol.numpy())
    else:
        out.update(out_volume_probe=vol.numpy()[:, ::16, ::3, ::3, ::3], in_volume_probe=vol_in.numpy()[:, ::16, ::3, ::3, ::3])
    path = os.path.join(REPO, "tests", "golden", f"ref_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, os.path.getsize(path) // 1024, "KiB",
          "wnf range", float(wnf.min()), float(wnf.max()))


def unet_case(name, G, B, seed):
    """Reference Abstract3DUNet as-is (components/unet3d.py) on a dense random volume."""
    from components.unet3d import Abstract3DUNet, DoubleConv
    hp = S.default_hparams(grid=G)
    sd = S.synthetic_state_dict(hp, seed=seed)
    net = Abstract3DUNet(in_channels=128, out_channels=128, final_sigmoid=False, basic_module=DoubleConv, f_maps=32,
                         layer_order="gcr", num_groups=8, num_levels=4, is_segmentation=False)
    pre = "unet_3d.abstract_3d_unet."
    net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    net.eval()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 128, G, G, G, generator=g)
    with torch.no_grad():
        y = net(x)
    path = os.path.join(REPO, "tests", "golden", f"ref_{name}.npz")
    np.savez_compressed(path, meta=np.array([G, B, seed], np.int64), y=y.numpy())
    print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


def grid_case():
    """VirtualGrid index maths (components/gridding.py) incl. the 64-bin -> G-cell LUTs and ArraySlicer."""
    from components.gridding import VirtualGrid, ArraySlicer
    out = {}
    bins = torch.arange(64).unsqueeze(1).repeat(1, 3)
    nocs = VirtualGrid(grid_shape=(64,) * 3, batch_size=1).idxs_to_points(bins)
    out["nocs_of_bin"] = nocs.numpy()
    for G in (8, 16, 32, 128):
        vg = VirtualGrid(grid_shape=(G,) * 3, batch_size=2)
        out[f"cell_of_bin_{G}"] = vg.get_points_grid_idxs(nocs).numpy()
        out[f"corner_of_bin_{G}"] = vg.idxs_to_points(vg.get_points_grid_idxs(nocs)).numpy()
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(500, 3, generator=g) * 1.2 - 0.1
    vg = VirtualGrid(grid_shape=(32, 32, 32), batch_size=2)
    bi = torch.randint(0, 2, (500,), generator=g)
    idx = vg.get_points_grid_idxs(pts, batch_idx=bi)
    out["rand_pts"] = pts.numpy()
    out["rand_batch"] = bi.numpy()
    out["rand_idx"] = idx.numpy()
    out["rand_flat"] = vg.flatten_idxs(idx).numpy()
    out["grid_points_5"] = VirtualGrid(grid_shape=(5,) * 3).get_grid_points(include_batch=False).numpy()
    for shape, chunks in (((128, 128, 128, 3), (64, 64, 64)), ((70, 64, 10, 3), (64, 64, 64))):
        sl = ArraySlicer(shape, chunks)
        out["slicer_%d" % shape[0]] = np.array([[(s.start, s.stop) for s in sl[i]] for i in range(len(sl))], np.int64)
    path = os.path.join(REPO, "tests", "golden", "ref_gridding.npz")
    np.savez_compressed(path, **out)
    print("gridding ->", path)


CONV_ORDERS = ("gcr", "cr", "crg", "cl", "ce", "bcr", "cbr", "cgr")


def conv_orders_case():
    """Reference SingleConv (components/unet3d.py:19-91, create_conv) for every layer order the docstrings name, on ONE shared set of parameters, and a
    small Abstract3DUNet with layer_order='crg' (the reference SingleConv's own default) end to end -> ref_conv_orders.npz (data only)."""
    from components.unet3d import Abstract3DUNet, DoubleConv, SingleConv
    g = torch.Generator().manual_seed(77)
    cin, cout, ng = 16, 32, 4
    rn = lambda *shape: torch.randn(*shape, generator=g)
    out = {"x": rn(2, cin, 4, 6, 8), "w": rn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5, "conv_bias": rn(cout) * 0.1}
    for tag, c in (("in", cin), ("out", cout)):
        out[f"gn_{tag}_weight"], out[f"gn_{tag}_bias"] = torch.rand(c, generator=g) + 0.5, rn(c) * 0.2
        out[f"bn_{tag}_weight"], out[f"bn_{tag}_bias"] = torch.rand(c, generator=g) + 0.5, rn(c) * 0.2
        out[f"bn_{tag}_mean"], out[f"bn_{tag}_var"] = rn(c) * 0.3, torch.rand(c, generator=g) + 0.5
    for order in CONV_ORDERS:
        m = SingleConv(cin, cout, kernel_size=3, order=order, num_groups=ng).eval()
        tag = "in" if any(ch in order[:order.index("c")] for ch in "gb") else "out"
        sd = {"conv.weight": out["w"]}
        if m.conv.bias is not None:
            sd["conv.bias"] = out["conv_bias"]
        if "g" in order:
            sd["groupnorm.weight"], sd["groupnorm.bias"] = out[f"gn_{tag}_weight"], out[f"gn_{tag}_bias"]
        if "b" in order:
            sd.update({"batchnorm.weight": out[f"bn_{tag}_weight"], "batchnorm.bias": out[f"bn_{tag}_bias"], "batchnorm.running_mean": out[f"bn_{tag}_mean"],
                       "batchnorm.running_var": out[f"bn_{tag}_var"], "batchnorm.num_batches_tracked": torch.tensor(0)})
        m.load_state_dict(sd)
        with torch.no_grad():
            out["y_" + order] = m(out["x"].clone())
    torch.manual_seed(78)
    net = Abstract3DUNet(in_channels=16, out_channels=16, final_sigmoid=False, basic_module=DoubleConv, f_maps=8, layer_order="crg", num_groups=4,
                         num_levels=2, is_segmentation=False).eval()
    with torch.no_grad():
        for prm in net.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.2 if prm.dim() > 1 else 0.3))
    xu = rn(2, 16, 8, 8, 8)
    with torch.no_grad():
        yu = net(xu)
    for k, v in net.state_dict().items():
        out["unet_crg." + k] = v
    out["unet_crg_x"], out["unet_crg_y"] = xu, yu
    path = os.path.join(REPO, "tests", "golden", "ref_conv_orders.npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("conv orders ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted: goldens can only be generated in the build container"
    torch.set_num_threads(8)
    install_stubs()
    if sys.argv[1:] == ["conv_orders"]:
        conv_orders_case()
        sys.exit(0)
    conv_orders_case()
    grid_case()
    unet_case("unet_g8", 8, 2, 5)
    unet_case("unet_g16", 16, 1, 6)
    run_case("small_max", B=2, n_points=512, grid=8, reduce_method="max", Q=16, seed=1)
    run_case("small_mean", B=2, n_points=640, grid=16, reduce_method="mean", Q=20, seed=2, store_full=False)
    run_case("dress_g32", B=1, n_points=6000, grid=32, reduce_method="max", Q=32, seed=0, store_full=False, stride=29)
