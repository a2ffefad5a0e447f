"""GPU (-m gpu): the API branches of the drop-in surface that predict.py / eval.py reach but the stage-by-stage parity tests do not:
the hole-prediction head (predict.py:202-209), ConvImplicitWNFPipeline.forward(data) with explicit query sets
(networks/conv_implicit_wnf.py:314-338), delete_invalid_verts on device tensors (common/marching_cubes_util.py:38-52), the sharded
(N > 1) data path run rank by rank on the one GPU, and a non-default device when the box has one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipeline as P  # noqa: E402
from garmentnets_amd import ops, parallel, synthetic as S  # noqa: E402
from garmentnets_amd.batch import Batch  # noqa: E402
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline  # noqa: E402
from garmentnets_amd.predict import predict_batch, to_host  # noqa: E402

DEV = "cuda:0"
TOL = 1e-4


def _model(hp, seed, dev=DEV, self_loops=True):
    m = ConvImplicitWNFPipeline(**hp)
    m.load_state_dict(S.synthetic_state_dict(hp, seed))
    m = m.to(dev).eval().requires_grad_(False)
    m.pointnet2_nocs.sa1_module.conv.add_self_loops = self_loops
    m.pointnet2_nocs.sa2_module.conv.add_self_loops = self_loops
    return m


def test_hole_prediction_branch_against_oracle():
    """predict_batch(use_hole_prediction=True): is_on_surface_logits = mc_surface_decoder at the mesh vertices, is_on_surface = logits > 0
    (predict.py:202-209) -- against the oracle's decoder on the HIP path's own feature volume and vertices; then delete_invalid_verts on
    the device result against the reference's numpy semantics"""
    hp = S.default_hparams(grid=32, reduce_method="max", mc_surface=True)
    sd = S.synthetic_state_dict(hp, 1)
    model = _model(hp, 1)
    assert model.mc_surface_decoder is not None
    x, pos, batch = S.synthetic_cloud(2, 3000, seed=8)
    data = Batch(sizes=[3000, 3000], x=x, pos=pos, batch=batch).to(DEV)
    res = predict_batch(model, data, volume_size=32, auto_level=True, use_hole_prediction=True)
    with torch.no_grad():
        u3 = model.unet3d_forward(model.pointnet2_forward(data))
        vol = u3["out_feature_volume"].cpu().contiguous()
    for b, r in enumerate(res):
        assert "is_on_surface_logits" in r and r["is_on_surface"].dtype == torch.bool
        q = r["verts"].float().cpu().view(1, -1, 3)
        ref = P.implicit_decoder(sd, "mc_surface_decoder", vol[b:b + 1], q).view(-1)
        np.testing.assert_allclose(r["is_on_surface_logits"].cpu().numpy(), ref.numpy(), rtol=0, atol=TOL)
        assert torch.equal(r["is_on_surface"], r["is_on_surface_logits"] > 0)
        host = to_host(r)
        assert host["is_on_surface"].dtype == np.bool_ and host["is_on_surface_logits"].dtype == np.float32
        # delete_invalid_verts with a mask that really removes something (random weights give one-sided logits)
        V = r["verts"].shape[0]
        mask = torch.rand(V, generator=torch.Generator().manual_seed(b)) > 0.3
        from garmentnets_amd.common.marching_cubes_util import delete_invalid_verts
        v_gpu, f_gpu = delete_invalid_verts(r["verts"], r["faces"], mask.to(DEV))
        verts, faces, ok = r["verts"].cpu().numpy(), r["faces"].cpu().numpy(), mask.numpy()
        valid = ok[faces[:, 0]] & ok[faces[:, 1]] & ok[faces[:, 2]]                     # common/marching_cubes_util.py:39-52 restated
        raw = faces[valid]
        used = np.unique(raw.flatten())
        remap = np.zeros(len(verts), dtype=faces.dtype)
        remap[used] = np.arange(len(used))
        assert f_gpu.dtype == r["faces"].dtype
        assert np.array_equal(v_gpu.cpu().numpy(), verts[used]) and np.array_equal(f_gpu.cpu().numpy(), remap[raw])


@pytest.mark.parametrize("V,F", [(0, 0), (5, 0), (1000, 3000), (70001, 140003)])
def test_mesh_compact_kernel(V, F):
    g = torch.Generator().manual_seed(V + F)
    verts = torch.randn(V, 3, generator=g, dtype=torch.float64)
    faces = torch.randint(0, max(V, 1), (F, 3), generator=g, dtype=torch.int32)
    ok = torch.rand(V, generator=g) > 0.2
    v, f = ops.mesh_compact(verts.to(DEV), faces.to(DEV), ok.to(DEV))
    fn, okn = faces.numpy(), ok.numpy()
    valid = okn[fn[:, 0]] & okn[fn[:, 1]] & okn[fn[:, 2]] if F else np.zeros(0, bool)
    used = np.unique(fn[valid].flatten())
    remap = np.zeros(V, dtype=np.int32)
    remap[used] = np.arange(len(used))
    assert np.array_equal(v.cpu().numpy(), verts.numpy()[used]) and np.array_equal(f.cpu().numpy(), remap[fn[valid]].reshape(-1, 3))
    v32, f32 = ops.mesh_compact(verts.float().to(DEV), faces.to(DEV), ok.to(DEV))
    assert np.array_equal(v32.cpu().numpy(), verts.float().numpy()[used]) and torch.equal(f32, f)


def test_forward_with_explicit_query_sets_against_oracle():
    """ConvImplicitWNFPipeline.forward(data) (networks/conv_implicit_wnf.py:314-338): data.volume_query_points / surf_query_points
    (B,M,3) -> volume_decoder_result / surface_decoder_result, the reference's result-dict layout, values vs the oracle decoders"""
    hp = S.default_hparams(grid=32, reduce_method="mean", mc_surface=True)
    sd = S.synthetic_state_dict(hp, 2)
    model = _model(hp, 2)
    x, pos, batch = S.synthetic_cloud(2, 2000, seed=3)
    g = torch.Generator().manual_seed(1)
    vq, sq, mq = torch.rand(2, 700, 3, generator=g), torch.rand(2, 333, 3, generator=g), torch.rand(2, 50, 3, generator=g)
    vq[0, 0], vq[0, 1] = 0.0, 1.0                                                  # exact borders
    data = Batch(sizes=[2000, 2000], x=x, pos=pos, batch=batch, volume_query_points=vq, surf_query_points=sq, mc_surf_query_points=mq).to(DEV)
    with torch.no_grad():
        out = model(data)
    assert set(out) == {"pointnet2_result", "unet3d_result", "volume_decoder_result", "surface_decoder_result", "mc_surface_decoder_result"}
    u3 = out["unet3d_result"]
    assert len(u3) == 1 and list(u3) == ["out_feature_volume"] and dict(u3)["out_feature_volume"].shape == (2, 128, 32, 32, 32)
    vol = u3["out_feature_volume"].cpu().contiguous()
    vd = out["volume_decoder_result"]
    assert vd["out_features"].shape == (2, 700, 1) and vd["pred_volume_value"].shape == (2, 700)
    np.testing.assert_allclose(vd["pred_volume_value"].cpu().numpy(), P.implicit_decoder(sd, "volume_decoder", vol, vq).squeeze(-1).numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["surface_decoder_result"]["out_features"].cpu().numpy(), P.implicit_decoder(sd, "surface_decoder", vol, sq).numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["mc_surface_decoder_result"]["out_features"].cpu().numpy(), P.implicit_decoder(sd, "mc_surface_decoder", vol, mq).numpy(),
                               rtol=0, atol=TOL)
    # and the whole chain against the oracle's own chain
    ref_p2 = P.pointnet2_forward(sd, hp, x, pos, batch)
    ref_vol = P.unet3d(sd, hp["unet3d_params"], P.volume_agg(sd, hp["volume_agg_params"], ref_p2["nocs_data"], 2))
    np.testing.assert_allclose(vd["pred_volume_value"].cpu().numpy(), P.implicit_decoder(sd, "volume_decoder", ref_vol, vq).squeeze(-1).numpy(), rtol=0, atol=TOL)


def test_sharded_data_path_rank_by_rank_equals_single_rank():
    """BASELINE config[3] in miniature, without the node: a seeded global batch of 4 garments.  (a) Each of 2 ranks' shards
    (parallel.shard_batch, what bench.py --gpus 2 runs) through predict_batch, one after the other on this GPU, equals the same two
    garments run as their own batch of two -- like for like, with PointConv's self-loop quirk ON (it makes a garment's features depend
    on its slot in the LOCAL batch).  (b) With the quirk off nothing depends on the slot: the concatenation of the two ranks' results
    is the single-rank batch-of-4 result, garment by garment (bit-equal WNF, faces, vertices)."""
    hp = S.default_hparams(grid=32, reduce_method="mean")
    total, n, seed, world = 4, 3000, 123, 2
    gx, gpos, gbatch = S.synthetic_cloud(total, n, seed=seed)
    shards = [parallel.shard_batch(total, n, seed, r, world) for r in range(world)]
    assert [s[1] for s in shards] == [(0, 2), (2, 4)]
    assert torch.equal(torch.cat([s[0].pos for s in shards]), gpos) and torch.equal(torch.cat([s[0].x for s in shards]), gx)
    model = _model(hp, 0)
    for sh, (lo, hi) in shards:
        res = predict_batch(model, sh.to(DEV), volume_size=32, auto_level=True)
        same = Batch(sizes=[n] * (hi - lo), x=gx[lo * n:hi * n], pos=gpos[lo * n:hi * n], batch=gbatch[lo * n:hi * n] - lo)
        ref = predict_batch(model, same.to(DEV), volume_size=32, auto_level=True)
        for a, b in zip(res, ref):
            assert torch.equal(a["wnf_volume"], b["wnf_volume"]) and torch.equal(a["faces"], b["faces"]) and torch.equal(a["verts"], b["verts"])
    model = _model(hp, 0, self_loops=False)
    whole = predict_batch(model, Batch(sizes=[n] * total, x=gx, pos=gpos, batch=gbatch).to(DEV), volume_size=32, auto_level=True)
    parts = []
    for sh, _ in shards:
        parts += predict_batch(model, sh.to(DEV), volume_size=32, auto_level=True)
    assert len(parts) == total
    for a, b in zip(parts, whole):
        assert torch.equal(a["wnf_volume"], b["wnf_volume"]) and torch.equal(a["faces"], b["faces"]) and torch.equal(a["verts"], b["verts"])
        assert torch.equal(a["pred_nocs"], b["pred_nocs"]) and torch.equal(a["warp_field"], b["warp_field"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU (main.gpu_id != 0 of the reference config)")
def test_non_default_device():
    """a model on cuda:1 while the process default device is 0: every gn_* call launches on cuda:1's stream (ops._stream / gn_stream)"""
    hp = S.default_hparams(grid=16)
    x, pos, batch = S.synthetic_cloud(1, 1500, seed=1)
    torch.cuda.set_device(0)
    out0 = predict_batch(_model(hp, 0, "cuda:0"), Batch(sizes=[1500], x=x, pos=pos, batch=batch).to("cuda:0"), volume_size=24, auto_level=True)[0]
    torch.cuda.set_device(0)
    out1 = predict_batch(_model(hp, 0, "cuda:1"), Batch(sizes=[1500], x=x, pos=pos, batch=batch).to("cuda:1"), volume_size=24, auto_level=True)[0]
    assert out1["wnf_volume"].device == torch.device("cuda:1")
    assert torch.equal(out0["wnf_volume"].cpu(), out1["wnf_volume"].cpu()) and torch.equal(out0["faces"].cpu(), out1["faces"].cpu())
    torch.cuda.set_device(0)


def test_predict_stream_equals_predict_batch():
    """predict.PredictJob / predict_stream (batch k+1's dense path queued before batch k's host-synchronising tail, tails on their own
    stream, two banks of iso slot buffers): every garment of every batch bit-equal to predict_batch run batch by batch -- including a
    batch whose meshes are empty at the fixed level (placeholder path) and the hole-prediction head"""
    from garmentnets_amd.predict import PredictJob, predict_stream
    hp = S.default_hparams(grid=32, reduce_method="mean", mc_surface=True)
    model = _model(hp, 3)
    batches = []
    for k in range(5):
        n = 2000 + 500 * (k % 2)
        x, pos, batch = S.synthetic_cloud(3, n, seed=40 + k)
        batches.append(Batch(sizes=[n] * 3, x=x, pos=pos, batch=batch).to(DEV))
    # a level every synthetic WNF straddles: the mid level of the first batch's first garment (fixed level: the pipelined path has no auto_level)
    probe = predict_batch(model, batches[0], volume_size=32, auto_level=True)
    w = probe[0]["wnf_volume"]
    level = 0.5 * (float(w.min()) + float(w.max()))
    ref = [predict_batch(model, b, volume_size=32, iso_surface_level=level, use_hole_prediction=True) for b in batches]
    got = list(predict_stream(model, batches, volume_size=32, iso_surface_level=level, use_hole_prediction=True))
    torch.cuda.synchronize()
    assert len(got) == len(ref)
    n_real = 0
    for rb, gb in zip(ref, got):
        assert len(rb) == len(gb)
        for r, g in zip(rb, gb):
            assert set(r) == set(g)
            for k in r:
                a, b = r[k], g[k]
                assert a.shape == b.shape and a.dtype == b.dtype, k
                assert torch.equal(torch.nan_to_num(a.double(), nan=-7.0), torch.nan_to_num(b.double(), nan=-7.0)), k
            n_real += int(not torch.isnan(r["verts"]).any())
    assert n_real >= 3
    # twice over the same batches (slot banks and the tail stream reused), interleaved with a plain predict_batch on the main stream
    j0 = PredictJob(model, batches[1], 32, level, bank=1)      # (predict_batch itself uses bank 0)
    j1 = PredictJob(model, batches[2], 32, level, bank=2)
    mid = predict_batch(model, batches[3], volume_size=32, iso_surface_level=level)
    r0, r1 = j0.finish(), j1.finish()
    for rb, gb in ((ref[1], r0), (ref[2], r1), (ref[3], mid)):
        for r, g in zip(rb, gb):
            assert torch.equal(r["faces"], g["faces"]) and torch.equal(torch.nan_to_num(r["verts"], nan=-7.0), torch.nan_to_num(g["verts"], nan=-7.0))
            assert torch.equal(torch.nan_to_num(r["warp_field"], nan=-7.0), torch.nan_to_num(g["warp_field"], nan=-7.0))
