"""GPU (-m gpu): the API branches of the drop-in surface that predict.py / eval.py reach but the stage-by-stage parity tests do not:
the hole-prediction head (predict.py:202-209), ConvImplicitWNFPipeline.forward(data) with explicit query sets
(networks/conv_implicit_wnf.py:314-338), delete_invalid_verts on device tensors (common/marching_cubes_util.py:38-52), the sharded
(N > 1) data path run rank by rank on the one GPU, and a non-default device when the box has one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipeline as P  # noqa: E402
from garmentnets_amd import arith as AR, ops, parallel, synthetic as S  # noqa: E402
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


@pytest.mark.parametrize("V,F,seed", [(7, 2, 0), (1000, 700, 1), (20000, 9000, 2), (300000, 250000, 3), (64, 640, 4)])
def test_mesh_largest_component_kernel(V, F, seed):
    """gn_mesh_largest_component against the oracle's restatement of igl.connected_components + np.argmax (eval.py:497-503): random triangle
    soups with many components (isolated vertices, ties between largest components), labels = lowest vertex of the component, run-to-run
    identical whatever the interleaving of the lock-free unions"""
    from oracle import mesh as OM
    rng = np.random.default_rng(seed)
    faces = rng.integers(0, V, size=(F, 3)).astype(np.int32)
    if seed == 0:
        faces = np.array([[2, 3, 4], [5, 0, 6]], dtype=np.int32)               # a tie: the component holding vertex 0 wins
    num, idx, sizes = OM.connected_components(faces, V)
    want = idx == np.argmax(sizes)
    first = np.full(num, V, dtype=np.int64)
    np.minimum.at(first, idx, np.arange(V))
    ft = torch.from_numpy(faces).to(DEV)
    mask, info = ops.mesh_largest_component(ft, V, with_labels=True)
    assert mask.dtype == torch.bool and np.array_equal(mask.cpu().numpy(), want)
    assert info["num_components"] == num and info["size"] == int(sizes.max()) and info["label"] == int(first[np.argmax(sizes)])
    assert np.array_equal(info["labels"].cpu().numpy().astype(np.int64), first[idx])
    for _ in range(3):                                                          # deterministic
        m2, i2 = ops.mesh_largest_component(ft.long(), V, with_labels=True)
        assert torch.equal(m2, mask) and torch.equal(i2["labels"], info["labels"])
    print(f"V={V} F={F}: {num} components, largest {int(sizes.max())}")


def test_mesh_largest_component_errors_and_hole_removal_on_a_real_mesh():
    """error contract (no faces: np.argmax of an empty sequence -> ValueError; a face index out of range -> IndexError as numpy) and the
    reference's whole hole removal (eval.py:529-548: threshold -> delete_invalid_verts -> largest component -> delete_invalid_verts) on a
    marching-cubes mesh with several pieces, device tensors against the oracle's numpy restatement"""
    from oracle import mesh as OM
    from garmentnets_amd.common.marching_cubes_util import marching_cubes, remove_holes, largest_connected_component
    with pytest.raises(ValueError):
        ops.mesh_largest_component(torch.zeros((0, 3), dtype=torch.int32, device=DEV), 5)
    with pytest.raises(IndexError):
        ops.mesh_largest_component(torch.tensor([[0, 1, 9]], dtype=torch.int32, device=DEV), 5)
    # three blobs of different size in a 48^3 volume
    z, y, x = np.meshgrid(*([np.arange(48, dtype=np.float32)] * 3), indexing="ij")
    vol = np.zeros((48, 48, 48), dtype=np.float32)
    for c, r in (((12, 12, 12), 7.5), ((32, 30, 28), 11.3), ((10, 38, 36), 5.2)):
        vol = np.maximum(vol, 1.0 - np.sqrt((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / r)
    verts, faces, _, values, _ = marching_cubes(torch.from_numpy(vol).to(DEV), 0.3)
    vn, fn = verts.cpu().numpy(), faces.cpu().numpy()
    assert OM.connected_components(fn, len(vn))[0] == 3
    is_cc = largest_connected_component(faces, verts.shape[0])
    assert np.array_equal(is_cc.cpu().numpy(), OM.largest_component_mask(fn, len(vn)))
    assert np.array_equal(largest_connected_component(faces.cpu(), verts.shape[0]).numpy(), is_cc.cpu().numpy()) and not largest_connected_component(faces.cpu(), 5 + verts.shape[0]).is_cuda
    # hole head stand-in: a per-vertex value that cuts a band out of the big blob (splitting it) and all of the smallest one
    pv = torch.from_numpy(((np.abs(vn[:, 0] - 32.0) > 2.0) & (vn[:, 1] < 34.0)).astype(np.float32)).to(DEV)
    extra = verts * 2.0 + 1.0
    v2, f2, e2 = remove_holes(verts, faces, pv, 0.5, extra_verts=(extra,))
    rv, rf = OM.remove_holes(vn, fn, pv.cpu().numpy(), 0.5)
    assert np.array_equal(v2.cpu().numpy(), rv) and np.array_equal(f2.cpu().numpy(), rf) and np.array_equal(e2.cpu().numpy(), rv * 2.0 + 1.0)
    assert 0 < len(rv) < len(vn) and OM.connected_components(rf, len(rv))[0] == 1
    print(f"hole removal: {len(vn)} -> {len(rv)} vertices, {len(fn)} -> {len(rf)} faces")


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
    stream, every job with its own iso-surface buffers): every garment of every batch bit-equal to predict_batch run batch by batch -- including a
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
    # the host path (SURVEY.md 8d's metric includes the D2H copy of every mesh): finish(host=True) copies on the tail stream -- same bytes
    got_host = list(predict_stream(model, batches, volume_size=32, iso_surface_level=level, use_hole_prediction=True, host=True))
    for rb, hb in zip(ref, got_host):
        for r, hres in zip(rb, hb):
            want = to_host(r)
            assert set(want) == set(hres)
            for k in want:
                assert isinstance(hres[k], np.ndarray) and want[k].dtype == hres[k].dtype
                assert np.array_equal(want[k], hres[k], equal_nan=want[k].dtype.kind == "f"), k
    # twice over the same batches (the tail stream reused), interleaved with a plain predict_batch on the main stream
    j0 = PredictJob(model, batches[1], 32, level)              # (every job owns its buffers)
    j1 = PredictJob(model, batches[2], 32, level)
    mid = predict_batch(model, batches[3], volume_size=32, iso_surface_level=level)
    r0, r1 = j0.finish(), j1.finish()
    for rb, gb in ((ref[1], r0), (ref[2], r1), (ref[3], mid)):
        for r, g in zip(rb, gb):
            assert torch.equal(r["faces"], g["faces"]) and torch.equal(torch.nan_to_num(r["verts"], nan=-7.0), torch.nan_to_num(g["verts"], nan=-7.0))
            assert torch.equal(torch.nan_to_num(r["warp_field"], nan=-7.0), torch.nan_to_num(g["warp_field"], nan=-7.0))


def test_forward_volume_task_space_against_oracle():
    """ConvImplicitWNFPipeline(volume_task_space=True).forward(data) (networks/conv_implicit_wnf.py:279-311,321-323): the gridding runs on
    the normalised simulation coordinates (scale / offset from data.cloth_sim_aabb, first sample's for the whole batch) instead of the
    predicted NOCS coordinates -- scale / offset against a restatement of :299-313, the gridded cells exactly, the decoders within 1e-4
    of the oracle chain fed the same positions"""
    hp = S.default_hparams(grid=32, reduce_method="mean")
    hp["volume_task_space"] = True
    sd = S.synthetic_state_dict(hp, 4)
    model = _model(hp, 4)
    assert model.volume_task_space
    x, pos, batch = S.synthetic_cloud(2, 2000, seed=9)
    aabb = torch.tensor([[[-0.32, -0.30, -0.85], [0.31, 0.33, 0.02]]], dtype=torch.float32).repeat(2, 1, 1)
    aabb[1] *= 1.3                                                 # (only the first sample's scale / offset is used, as in the reference)
    g = torch.Generator().manual_seed(2)
    vq, sq = torch.rand(2, 500, 3, generator=g), torch.rand(2, 200, 3, generator=g)
    data = Batch(sizes=[2000, 2000], x=x, pos=pos, batch=batch, volume_query_points=vq, surf_query_points=sq, cloth_sim_aabb=aabb).to(DEV)
    with torch.no_grad():
        out = model(data)
    # conv_implicit_wnf.py:299-313 restated on the host
    nr = 0.45
    radius = aabb.abs().max(dim=1)[0][:, :2]
    scale = torch.minimum((nr / radius).min(dim=1)[0], (2 * nr) / (aabb[:, 1, 2] - aabb[:, 0, 2]))
    offset = torch.full((2, 3), 0.5)
    offset[:, 2] = 1 - 0.05 - aabb[:, 1, 2] * scale
    sc, of = model.get_aabb_scale_offset(aabb)
    assert torch.equal(sc, scale) and torch.equal(of, offset)
    new_pos = pos * scale[0] + offset[0]
    nd = out["pointnet2_result"]["nocs_data"]
    assert torch.equal(nd.pos.cpu(), new_pos) and float(new_pos.min()) >= 0.0 and float(new_pos.max()) <= 1.0
    ref_p2 = P.pointnet2_forward(sd, hp, x, pos, batch)
    ref_nd = dict(ref_p2["nocs_data"])
    ref_nd["pos"] = new_pos
    ref_vin = P.volume_agg(sd, hp["volume_agg_params"], ref_nd, 2)
    with torch.no_grad():
        vin = model.volume_agg(nd)
    assert torch.equal(vin.cpu() != 0, ref_vin != 0)
    ref_vol = P.unet3d(sd, hp["unet3d_params"], ref_vin)
    np.testing.assert_allclose(out["volume_decoder_result"]["pred_volume_value"].cpu().numpy(),
                               P.implicit_decoder(sd, "volume_decoder", ref_vol, vq).squeeze(-1).numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(out["surface_decoder_result"]["out_features"].cpu().numpy(), P.implicit_decoder(sd, "surface_decoder", ref_vol, sq).numpy(),
                               rtol=0, atol=TOL)


def test_surface_decoder_nan_triggers_the_fp32_rerun(monkeypatch):
    """the surface / hole decoders run the split-operand kernel with their OWN weights and scales: a NaN there (finite WNF) must also
    send the batch to the fp32 kernels -- predict_batch and PredictJob.finish alike"""
    from garmentnets_amd import predict as PR
    hp = S.default_hparams(grid=16, reduce_method="max", mc_surface=True)
    model = _model(hp, 3)
    x, pos, batch = S.synthetic_cloud(2, 1500, seed=5)
    data = Batch(sizes=[1500, 1500], x=x, pos=pos, batch=batch).to(DEV)
    probe = predict_batch(model, data, volume_size=24, auto_level=True)
    w = probe[0]["wnf_volume"]
    level = 0.5 * (float(w.min()) + float(w.max()))
    want = predict_batch(model, data, volume_size=24, iso_surface_level=level, use_hole_prediction=True, arith=model.arith.strict_fp32())
    assert not any(bool(torch.isnan(r["verts"]).any()) for r in want)
    orig = ops.implicit_decode_split
    for which in (3, 1):                                   # poison the warp-field decoder (3 outputs), then the hole decoder (1 output, surface queries)
        state = {"armed": True}

        def poisoned(xin, pack, out=None, xscale=None, _which=which):
            r = orig(xin, pack, out=out, xscale=xscale)
            if state["armed"] and pack.out_channels == _which and r.shape[0] != 24 * 24 * 24 and r.shape[0] > 0:
                r[0] = float("nan")
                state["armed"] = False
            return r
        orig_b = ops.implicit_decode_split_batch

        def poisoned_batch(xin, pack, out, xscale=None, _which=which):       # (round 6: a batch's surface queries go through the batched entry)
            r = orig_b(xin, pack, out, xscale=xscale)
            if state["armed"] and pack.out_channels == _which:
                r[0, 0] = float("nan")
                state["armed"] = False
            return r
        monkeypatch.setattr(ops, "implicit_decode_split", poisoned)
        monkeypatch.setattr(ops, "implicit_decode_split_batch", poisoned_batch)
        for runner in ("batch", "job"):
            state["armed"] = True
            PR._FALLBACKS["count"] = 0
            with pytest.warns(UserWarning):
                if runner == "batch":
                    got = predict_batch(model, data, volume_size=24, iso_surface_level=level, use_hole_prediction=True)
                else:
                    got = PR.PredictJob(model, data, 24, level, use_hole_prediction=True).finish()
            assert PR._FALLBACKS["count"] == 1 and not state["armed"], (which, runner)
            for a, b in zip(got, want):
                assert torch.equal(a["warp_field"], b["warp_field"]) and torch.equal(a["is_on_surface_logits"], b["is_on_surface_logits"])
                assert torch.equal(a["wnf_volume"], b["wnf_volume"]) and torch.equal(a["faces"], b["faces"])
        monkeypatch.setattr(ops, "implicit_decode_split", orig)
        monkeypatch.setattr(ops, "implicit_decode_split_batch", orig_b)


def test_decoder_input_scale_is_not_keyed_on_a_recycled_address():
    """ImplicitWNFDecoder called directly on materialised volumes (the reference's literal path): batch k+1's volume usually lands at the
    address batch k's was freed from, with the same shape and version -- its fp16 input scale must come from ITS statistics.  The second
    volume is 2^20 times larger: with the first one's scale the split kernel overflows fp16 (NaN) or loses the second plane."""
    from garmentnets_amd.networks.conv_implicit_wnf import ImplicitWNFDecoder
    dec = ImplicitWNFDecoder((128, 256, 256, 1), batch_norm=True)
    dec.load_state_dict({k: S.synthetic_tensor("volume_decoder." + k, tuple(v.shape), seed=3) for k, v in dec.state_dict().items()})
    dec = dec.to(DEV).eval()
    g = torch.Generator().manual_seed(0)
    q = torch.rand(1, 4096, 3, generator=g).to(DEV)
    base = torch.randn(1, 128, 8, 8, 8, generator=g)
    ptrs, errs = [], []
    for k, gain in enumerate((1.0, 2.0 ** 20, 2.0 ** -20)):
        vol = (base * gain).to(DEV)
        ptrs.append(vol.data_ptr())
        with torch.no_grad():
            out = dec(vol, q)
            ref = dec(vol, q, arith=AR.DEFAULT.strict_fp32())
        assert bool(torch.isfinite(out).all())
        errs.append(float((out - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
        del vol, out, ref
    print(f"decoder on recycled volume addresses {[hex(p) for p in ptrs]}: relative error vs the fp32 kernel {errs}")
    assert max(errs) <= 2e-5


def _write_synthetic_dataset(path, n_samples, rng):
    """a garmentnets dataset store in the reference's layout (datasets/conv_implicit_wnf_dataset.py:134-181 reads it): per sample
    point_cloud/{point,nocs,rgb,sizes}, mesh/{cloth_verts,cloth_nocs_verts,cloth_faces_tri}, marching_cube_mesh/{marching_cube_verts,
    marching_cube_faces,is_vertex_on_surface} + attrs; summary/cloth_aabb_union"""
    from garmentnets_amd.io import zarr_store
    root = zarr_store.open_group(path)
    root.require_group("summary").array("cloth_aabb_union", np.array([[-0.4, -0.4, -0.9], [0.4, 0.4, 0.05]], dtype=np.float32))
    keys = []
    for i in range(n_samples):
        key = f"{i:05d}_Dress_{i:06d}_0"
        keys.append(key)
        sg = root.require_group("samples").require_group(key)
        sg.put_attrs({"scale": 1.0 + 0.1 * i, "gender": i % 2, "sample_id": f"{i:05d}_Dress", "garment_name": "Dress", "grip_vertex_idx": 3 + i})
        x, pos, _ = S.synthetic_cloud(1, 2400, seed=70 + i)
        pos = pos.numpy()
        nocs = ((pos - pos.min(0)) / (pos.max(0) - pos.min(0))).astype(np.float32)
        pc, mesh, mc = sg.require_group("point_cloud"), sg.require_group("mesh"), sg.require_group("marching_cube_mesh")
        pc.array("point", pos, chunks=(1000, 3), compressor=("zlib", 1))
        pc.array("nocs", nocs)
        pc.array("rgb", (x.numpy() * 255).astype(np.uint8))
        pc.array("sizes", np.array([600, 600, 600, 600], dtype=np.int64))
        mesh.array("cloth_verts", pos[:300].astype(np.float32))
        mesh.array("cloth_nocs_verts", nocs[:300])
        mesh.array("cloth_faces_tri", rng.integers(0, 300, (500, 3)).astype(np.int32))
        mc.array("marching_cube_verts", rng.random((700, 3)).astype(np.float32), chunks=(256, 3), compressor=("zlib", 1))
        mc.array("marching_cube_faces", rng.integers(0, 700, (1300, 3)).astype(np.int32))
        mc.array("is_vertex_on_surface", rng.random(700) > 0.4)
    return keys


def _zarr_v2_read(store, path):
    """independent minimal Zarr v2 reader (spec only; shares no code with garmentnets_amd.io.zarr_store)"""
    import itertools
    import zlib
    base = os.path.join(store, path)
    meta = json.load(open(os.path.join(base, ".zarray")))
    assert meta["zarr_format"] == 2 and meta["order"] == "C" and not meta.get("filters")
    shape, chunks, dt = meta["shape"], meta["chunks"], np.dtype(meta["dtype"])
    out = np.full(shape, meta["fill_value"] if meta["fill_value"] is not None else 0, dtype=dt)
    for idx in itertools.product(*[range(max(1, -(-s // c))) for s, c in zip(shape, chunks)]):
        f = os.path.join(base, ".".join(str(i) for i in idx) if idx else "0")
        if not os.path.exists(f):
            continue
        raw = open(f, "rb").read()
        if meta["compressor"] is not None:
            assert meta["compressor"]["id"] == "zlib"
            raw = zlib.decompress(raw)
        chunk = np.frombuffer(raw, dtype=dt).reshape(chunks)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out


def test_predict_main_writes_the_store_eval_reads(tmp_path):
    """f1 end to end: predict.main over a small dataset store (rotation augmentation ON as in predict_default.yaml, static seed) ->
    prediction.zarr, read back with an independent Zarr v2 reader along EVERY key eval.py touches (eval.py:58-70,106-110,147-152,
    193-208): marching_cubes_mesh/*, point_cloud/{gt_nocs,pred_nocs,...}, misc/{gt_nocs_grip_point,...}, gt_marching_cubes_mesh/* (copy of
    the input sample's marching_cube_mesh), gt_mesh/* (cloth_verts rotated by the sample's augmentation matrix), per-sample attrs
    (predict.py:120-136) and the root `subset` attr -- and against predict_batch on the same dataset items"""
    from garmentnets_amd import predict as PR
    from garmentnets_amd.io.dataset import GarmentInputDataset
    rng = np.random.default_rng(5)
    din, dout = str(tmp_path / "garmentnets_dataset.zarr"), str(tmp_path / "prediction.zarr")
    keys = _write_synthetic_dataset(din, 2, rng)
    PR.main(["--zarr_in", din, "--zarr_out", dout, "--num_samples", "2", "--num_pc_sample", "1800", "--num_views", "3", "--grid", "16", "--volume_size", "24",
             "--auto_level", "--static_epoch_seed", "--random_rot_range", "-180", "180", "--subset", "all"])
    assert json.load(open(os.path.join(dout, ".zattrs")))["subset"] == "all"
    ds = GarmentInputDataset(din, num_pc_sample=1800, num_views=3, static_epoch_seed=True, enable_augumentation=True, random_rot_range=(-180, 180))
    hp = S.default_hparams(grid=16, reduce_method="max")
    model = _model(hp, 0)
    for i, key in enumerate(keys):
        base = os.path.join("samples", key)
        attrs = json.load(open(os.path.join(dout, base, ".zattrs")))
        assert attrs == {"scale": 1.0 + 0.1 * i, "gender": i % 2, "sample_id": f"{i:05d}_Dress", "garment_name": "Dress", "grip_vertex_idx": 3 + i, "batch_idx": i}
        item = ds[i]
        data = GarmentInputDataset.collate([item])
        rot = item["input_aug_rot_mat"][0]
        assert not np.allclose(rot, np.eye(3))
        res = PR.to_host(predict_batch(model, data.to(DEV), volume_size=24, auto_level=True)[0])
        rd = lambda *p: _zarr_v2_read(dout, os.path.join(base, *p))
        # eval.py:67-70,195-198: the predicted mesh
        for k in ("verts", "faces", "normals", "volume_value", "volume_gradient_magnitude", "warp_field"):
            got = rd("marching_cubes_mesh", k)
            assert got.dtype == res[k].dtype and np.array_equal(got, res[k]), k
        assert rd("marching_cubes_mesh", "verts").shape[0] > 50
        # eval.py:108-110, predict.py:220-227
        assert np.array_equal(rd("point_cloud", "gt_nocs"), item["y"]) and rd("point_cloud", "gt_nocs").dtype == np.float32
        assert rd("point_cloud", "pred_nocs").shape == (1800, 3) and rd("point_cloud", "pred_nocs_logits").shape == (1800, 192)
        assert np.array_equal(rd("point_cloud", "input_points"), item["pos"]) and rd("point_cloud", "input_rgb").dtype == np.uint8
        assert rd("point_cloud", "pred_nocs_confidence").shape == (1800, 3)
        # eval.py:149-152, predict.py:268-274
        assert np.array_equal(rd("misc", "gt_nocs_grip_point"), item["nocs_grip_point"][0])
        for k, shp in (("pred_nocs_grip_point", (3,)), ("pred_global_nocs_grip_point", (3,)), ("pred_global_confidence", (64, 3)), ("global_feature", (1024,))):
            assert rd("misc", k).shape == shp, k
        # eval.py:62-65,205-208, predict.py:236-239: copy of the input group
        for k in ("marching_cube_verts", "marching_cube_faces", "is_vertex_on_surface"):
            src = _zarr_v2_read(din, os.path.join(base, "marching_cube_mesh", k))
            got = rd("gt_marching_cubes_mesh", k)
            assert got.dtype == src.dtype and np.array_equal(got, src), k
        # eval.py:200-203, predict.py:241-250: gt_mesh with the augmentation rotation on cloth_verts
        cv = _zarr_v2_read(din, os.path.join(base, "mesh", "cloth_verts"))
        assert np.array_equal(rd("gt_mesh", "cloth_verts"), cv @ rot.T)
        for k in ("cloth_nocs_verts", "cloth_faces_tri"):
            assert np.array_equal(rd("gt_mesh", k), _zarr_v2_read(din, os.path.join(base, "mesh", k))), k


def test_self_loop_scope_example_gives_every_garment_its_batch_of_one_result():
    """PointConv's bipartite self-loop rule scoped per example (gn_sa_fused_scoped / gn_sa_gather_scoped, PointNet2NOCS.set_self_loop_scope):
    a ragged batch of four garments through the HIP path == the ORACLE run on each garment alone (batch of one: what predict.py, which
    asserts batch_size == 1, computes) -- and bit-equal to the HIP path's own batch-of-one runs; with the default "batch" scope the slots
    behind the first legitimately differ"""
    hp = S.default_hparams(grid=16, reduce_method="max")
    sd = S.synthetic_state_dict(hp, 11)
    sizes = [1500, 2100, 1800, 1500]
    clouds = [S.synthetic_cloud(1, n, seed=300 + i) for i, n in enumerate(sizes)]
    x, pos = torch.cat([c[0] for c in clouds]), torch.cat([c[1] for c in clouds])
    batch = torch.repeat_interleave(torch.arange(4), torch.tensor(sizes))
    data = Batch(sizes=sizes, x=x, pos=pos, batch=batch).to(DEV)
    for fused in (True, False):
        import garmentnets_amd.components.pointnet2 as CP
        saved = CP.FUSED_SA
        CP.FUSED_SA = fused
        try:
            model = _model(hp, 11)
            model.pointnet2_nocs.set_self_loop_scope("example")
            with torch.no_grad():
                got = model.pointnet2_forward(data)
                lo = 0
                for i, n in enumerate(sizes):
                    cx, cp, cb = clouds[i]
                    ref = P.pointnet2_forward(sd, hp, cx, cp, cb)
                    one = model.pointnet2_forward(Batch(sizes=[n], x=cx, pos=cp, batch=cb).to(DEV))
                    sl = slice(lo, lo + n)
                    assert torch.equal(got["per_point_logits"][sl], one["per_point_logits"]), (fused, i)
                    assert torch.equal(got["global_feature"][i], one["global_feature"][0]), (fused, i)
                    assert float((got["per_point_logits"][sl].cpu() - ref["per_point_logits"]).abs().max()) <= TOL
                    assert float((got["global_feature"][i].cpu() - ref["global_feature"][0]).abs().max()) <= TOL
                    lo += n
                model.pointnet2_nocs.set_self_loop_scope("batch")
                lit = model.pointnet2_forward(data)
                assert torch.equal(lit["per_point_logits"][:sizes[0]], got["per_point_logits"][:sizes[0]])     # slot 0: the same rule
                assert not torch.equal(lit["per_point_logits"][sizes[0]:], got["per_point_logits"][sizes[0]:])
        finally:
            CP.FUSED_SA = saved
    with pytest.raises(ValueError):
        model.pointnet2_nocs.set_self_loop_scope("garment")


def test_two_host_threads_two_jobs_two_arithmetics():
    """reentrant across host threads -- no module-level mutable state on the call path (ops' per-call device is
    thread-local, the CSR tables sit behind a lock, every thread has its own tail stream).  Two Python threads, each on its own stream,
    drive a PredictJob with a DIFFERENT Arith on different batches at the same time, several rounds; every result must equal the
    sequential one bit for bit"""
    import threading
    from garmentnets_amd.predict import PredictJob
    hp = S.default_hparams(grid=32, reduce_method="mean")
    model = _model(hp, 5)
    ariths = [AR.Arith.named("f16x2", "f16x2"), AR.Arith.named("fp32", "fp32", sparse_first_conv=False)]
    batches = []
    for k in range(2):
        n = 1800 + 400 * k
        x, pos, batch = S.synthetic_cloud(2, n, seed=500 + k)
        batches.append(Batch(sizes=[n] * 2, x=x, pos=pos, batch=batch).to(DEV))
    w = predict_batch(model, batches[0], volume_size=32, auto_level=True)[0]["wnf_volume"]
    level = 0.5 * (float(w.min()) + float(w.max()))
    ref = [predict_batch(model, b, volume_size=32, iso_surface_level=level, arith=a) for b, a in zip(batches, ariths)]
    torch.cuda.synchronize()
    assert not torch.equal(ref[0][0]["wnf_volume"], predict_batch(model, batches[0], volume_size=32, iso_surface_level=level, arith=ariths[1])[0]["wnf_volume"])
    errors, results = [], [None, None]
    gate = threading.Barrier(2)

    def work(t):
        try:
            stream = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(stream):
                for _ in range(6):
                    gate.wait(timeout=120)
                    job = PredictJob(model, batches[t], 32, level, arith=ariths[t])
                    out = job.finish(host=True)
                results[t] = out
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))
            gate.abort()

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in threads]
    [t.join(timeout=600) for t in threads]
    assert not errors, errors
    for t in range(2):
        for r, g in zip(ref[t], results[t]):
            host = to_host(r)
            assert set(host) == set(g)
            for k in host:
                assert host[k].dtype == g[k].dtype and np.array_equal(host[k], g[k], equal_nan=host[k].dtype.kind == "f"), (t, k)


def test_predict_main_subset_and_batched_cli(tmp_path):
    """predict.main as a user runs it on a store: --subset test iterates exactly the data module's seeded test split (static_epoch_seed as
    the reference's val_dataset), and --batch_size 4 (two batches in flight, per-example self-loop scope) writes, for every sample, what the
    reference-shaped batch_size 1 loop writes"""
    from garmentnets_amd import predict as PR
    from garmentnets_amd.io.dataset import GarmentInputDataset, instance_split
    rng = np.random.default_rng(9)
    din = str(tmp_path / "ds.zarr")
    keys = _write_synthetic_dataset(din, 10, rng)
    # a fixed level the synthetic checkpoint's WNF volumes straddle (the pipelined path has no auto_level): mid level of sample 0
    ds0 = GarmentInputDataset(din, num_pc_sample=1500, num_views=3, static_epoch_seed=True, random_rot_range=(-180, 180))
    w = predict_batch(_model(S.default_hparams(grid=16, reduce_method="max"), 0), GarmentInputDataset.collate([ds0[0]]).to(DEV), volume_size=24, auto_level=True)[0]["wnf_volume"]
    level = 0.5 * (float(w.min()) + float(w.max()))
    common = ["--zarr_in", din, "--num_pc_sample", "1500", "--num_views", "3", "--grid", "16", "--volume_size", "24", "--iso_surface_level", repr(level)]
    PR.main(common + ["--zarr_out", str(tmp_path / "p_test.zarr"), "--subset", "test"])
    ids = [f"{i:05d}_Dress" for i in range(10)]
    want = instance_split(ids, (8, 1, 1), 0)["test"]
    assert len(want) == 1 and sorted(d for d in os.listdir(tmp_path / "p_test.zarr" / "samples") if not d.startswith(".")) == [keys[int(i)] for i in want]
    ds = GarmentInputDataset(din, num_pc_sample=1500, num_views=3, static_epoch_seed=True, random_rot_range=(-180, 180))
    item = ds[int(want[0])]
    got_pts = _zarr_v2_read(str(tmp_path / "p_test.zarr"), os.path.join("samples", keys[int(want[0])], "point_cloud", "input_points"))
    assert np.array_equal(got_pts, item["pos"])
    # batched CLI vs the batch-of-one loop over the same six samples
    six = common + ["--subset", "all", "--static_epoch_seed", "--num_samples", "6"]
    PR.main(six + ["--zarr_out", str(tmp_path / "p_b1.zarr"), "--batch_size", "1", "--in_flight", "1"])
    PR.main(six + ["--zarr_out", str(tmp_path / "p_b4.zarr"), "--batch_size", "4"])
    n_real = 0
    for i, key in enumerate(keys[:6]):
        a = lambda *q: _zarr_v2_read(str(tmp_path / "p_b1.zarr"), os.path.join("samples", key, *q))
        b = lambda *q: _zarr_v2_read(str(tmp_path / "p_b4.zarr"), os.path.join("samples", key, *q))
        for k in ("pred_nocs", "pred_nocs_logits", "pred_nocs_confidence", "input_points", "input_rgb", "gt_nocs"):
            assert np.array_equal(a("point_cloud", k), b("point_cloud", k)), (key, k)
        for k in ("global_feature", "pred_nocs_grip_point", "pred_global_nocs_grip_point", "gt_nocs_grip_point"):
            assert np.array_equal(a("misc", k), b("misc", k)), (key, k)
        va, vb = a("marching_cubes_mesh", "verts"), b("marching_cubes_mesh", "verts")
        assert va.shape == vb.shape and np.array_equal(a("marching_cubes_mesh", "faces"), b("marching_cubes_mesh", "faces")), key
        assert np.allclose(va, vb, atol=1e-5, equal_nan=True) and np.allclose(a("marching_cubes_mesh", "warp_field"), b("marching_cubes_mesh", "warp_field"), atol=1e-5, equal_nan=True)
        assert json.load(open(tmp_path / "p_b4.zarr" / "samples" / key / ".zattrs"))["batch_idx"] == i
        n_real += int(not np.isnan(va).any())
    assert n_real >= 3


def _two_rank_bench(backend):
    """the N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per rank), on a box with ONE GPU: two
    ranks share cuda:0 and exchange their metrics over gloo (GARMENTNETS_DIST_BACKEND=gloo; RCCL needs one device per rank).  The line
    must see both ranks, the global batch must be the two shards, and the per-garment result checksums, in rank order, must equal what
    each shard gives in THIS process (like for like: PointConv's self-loop quirk ties a garment's result to its slot in the LOCAL batch)"""
    import bench
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(batch=2, points=2000, grid=32, reduce="mean", Q=32)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GARMENTNETS_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", str(cfg["batch"]), "--points", str(cfg["points"]),
           "--grid", str(cfg["grid"]), "--volume-size", str(cfg["Q"]), "--no-strict-pass", "--no-host-io-pass", "--no-occupancy-pass", "--no-in-flight-pass",
           "--no-validate", "--no-cpu-baseline", "--no-pmc"]
    out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 prints, rank 1 does not
    compact = json.loads(lines[0])
    assert len(lines[0]) < bench.COMPACT_LIMIT and tuple(compact) == bench.COMPACT_KEYS          # the stdout contract (tests/test_bench_line.py)
    assert compact["n_gpus"] == 2 and compact["rccl_ranks_seen"] == 2 and compact["dist_backend"] == backend and compact["roofline"]["kernel"]
    assert len(compact["per_rank"]["garments_per_s"]) == 2 and compact["per_rank"]["slowest_over_fastest"] >= 1.0
    # everything else travels in the detail dict: rank 0's stderr (and the file the line names)
    det = [l for l in out.stderr.splitlines() if l.startswith("[bench detail] ")]
    assert len(det) == 1
    line = json.loads(det[0][len("[bench detail] "):])
    assert line["value"] == pytest.approx(compact["value"], rel=1e-5)
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and line["dist_backend"] == backend
    assert len(line["per_rank"]["seconds"]) == 2 and all(t > 0 for t in line["per_rank"]["seconds"]) and line["per_rank"]["slowest_over_fastest"] >= 1.0
    assert line["per_rank"]["host_affinity_rank0"]["pinned"], line["per_rank"]
    assert line["config"]["global_batch"] == 4 and line["config"]["batch_per_gpu"] == 2 and line["value"] > 0
    assert abs(line["value"] - 4 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-6 * line["value"]
    sums = line["garment_checksums"]
    assert len(sums) == 4
    want = []
    for r in range(2):
        hp, sd, shard, (lo, hi) = bench.bench_inputs(cfg["batch"], cfg["points"], cfg["grid"], cfg["reduce"], "planted", r, 2)
        assert (lo, hi) == (2 * r, 2 * r + 2)
        model = ConvImplicitWNFPipeline(**hp)
        model.load_state_dict(sd)
        dev_r = DEV if backend == "gloo" else f"cuda:{r}"          # nccl: rank r computed on device r -- so does the reference run here
        model = model.to(dev_r).eval().requires_grad_(False)
        model.arith = model.arith.replace(sparse_first_conv=False)
        res = predict_batch(model, shard.to(dev_r), volume_size=cfg["Q"], iso_surface_level=0.5, auto_level=line["config"]["iso_level"] != 0.5)
        want += [float(x["wnf_volume"].double().sum()) for x in res]
    assert np.allclose(sums, want, rtol=1e-6, atol=1e-6), (sums, want)
    assert len(set(round(v, 3) for v in sums)) == 4                 # four different garments


def test_two_process_bench_on_one_gpu():
    """two ranks share cuda:0 and exchange their metrics over gloo (RCCL needs one device per rank) -- runs on the 1-GPU box"""
    _two_rank_bench("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs the day the box has two GPUs")
def test_two_process_bench_over_rccl():
    """the same launch with the production backend: init_process_group("nccl", device_id=cuda:LOCAL_RANK), rank 1's whole data path on
    device ordinal 1 (gn_stream's device binding), the metrics all-gather over RCCL; checksums against in-process runs on the same devices"""
    _two_rank_bench("nccl")
