"""GPU (-m gpu): the HIP path, called through the C ABI, against the pinned CPU oracle and the golden vectors.

Bit-exact for index / integer results (fps order, ball-query tables, cell indices, NOCS bins, marching-cubes faces and
vertex ids, GGM); fp32 results within the tolerance written next to each check (north_star: 1e-4 on WNF / NOCS).
"""
import hashlib
import warnings
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from oracle import pipeline as P  # noqa: E402
from garmentnets_amd import arith as AR, ops, synthetic as S  # noqa: E402
from garmentnets_amd.batch import Batch  # noqa: E402
from garmentnets_amd.common import marching_cubes_util as MCU  # noqa: E402
from garmentnets_amd.components.pointnet2 import Segments  # noqa: E402
from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline  # noqa: E402

DEV = "cuda:0"
TOL = 1e-4   # north_star tolerance for fp32 WNF / NOCS / features


def _sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def _ragged_cloud(sizes, seed):
    xs, ps, bs = [], [], []
    for b, n in enumerate(sizes):
        x, p, _ = S.synthetic_cloud(1, n, seed=seed + b)
        xs.append(x); ps.append(p); bs.append(torch.full((n,), b, dtype=torch.int64))
    return torch.cat(xs), torch.cat(ps), torch.cat(bs)


def _model(hp, seed):
    m = ConvImplicitWNFPipeline(**hp)
    m.load_state_dict(S.synthetic_state_dict(hp, seed))
    return m.to(DEV).eval().requires_grad_(False)


# ------------------------------------------------------------------------------------------------ point ops
@pytest.mark.parametrize("sizes,ratio", [([6000, 6000], 0.5), ([3000], 0.25), ([700, 1, 333, 64, 65], 0.5), ([9000], 0.25),
                                         ([6145, 6144], 0.25), ([8192, 8193], 0.125), ([13000], 0.05),
                                         ([20000, 300], 0.1), ([36864], 0.02),        # > 16384 points: the LDS-resident kernel (round 6)
                                         # 1025 .. 8192 points: the spatially pruned kernel (round 6) -- ragged batches, every block count
                                         ([6000, 5, 1, 1500, 64], 0.5), ([8192, 1025], 0.25), ([2048, 4100], 0.5), ([5000, 7168, 3073], 0.1)])
def test_fps_bit_exact(sizes, ratio):
    _, pos, batch = _ragged_cloud(sizes, 3)
    ptr = O.batch_to_ptr(batch.numpy())
    ref, optr = O.fps(pos.numpy(), ptr, ratio)
    seg = Segments(sizes, DEV)
    cseg = Segments([ops.fps_count(n, ratio) for n in sizes], DEV)
    assert list(cseg.ptr.cpu().numpy()) == list(optr)
    idx = ops.fps(pos.to(DEV), seg.ptr, cseg.ptr, max(sizes), cseg.total)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref)


@pytest.mark.parametrize("sizes", [[2000, 700], [6500], [64, 5]])
def test_fps_ties_keep_the_lowest_index(sizes):
    """points on a coarse integer lattice, many of them duplicated: most steps see exact ties of the running distance inside a lane, inside a
    wave and across waves (the ballot short cut's fallback, the DPP min, the cross-wave min) -- the oracle's rule is the lowest index"""
    g = torch.Generator().manual_seed(7)
    pos = torch.cat([torch.randint(0, 5, (n, 3), generator=g).float() * 0.25 for n in sizes])
    batch = torch.cat([torch.full((n,), b, dtype=torch.int64) for b, n in enumerate(sizes)])
    ptr = O.batch_to_ptr(batch.numpy())
    ref, optr = O.fps(pos.numpy(), ptr, 0.5)
    seg = Segments(sizes, DEV)
    cseg = Segments([ops.fps_count(n, 0.5) for n in sizes], DEV)
    idx = ops.fps(pos.to(DEV), seg.ptr, cseg.ptr, max(sizes), cseg.total)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref)


@pytest.mark.parametrize("n", [6000, 2500, 8000])
def test_fps_pruned_kernel_on_structured_clouds(n):
    """the spatially pruned kernel (fps_region_kernel: blocks whose bounding box the new sample cannot reach are skipped) on clouds where the skip
    rule has the most to decide: a thin shell (a surface, like a garment), tight clusters far apart (most blocks skipped from the first steps on),
    one cluster of exact duplicates (zero-extent boxes, running maxima that reach 0), points on a line (degenerate bounding box), and a flat sheet
    with lattice ties -- every index list is the oracle's"""
    g = torch.Generator().manual_seed(n)
    u = torch.randn(n, 3, generator=g)
    shell = u / u.norm(dim=1, keepdim=True) * (0.4 + 0.002 * torch.rand(n, 1, generator=g))
    centres = torch.rand(12, 3, generator=g) * 10
    clusters = centres[torch.randint(0, 12, (n,), generator=g)] + 1e-3 * torch.randn(n, 3, generator=g)
    dup = clusters.clone()
    dup[: 2 * n // 3] = dup[0]
    line = torch.zeros(n, 3)
    line[:, 1] = torch.rand(n, generator=g)
    sheet = torch.cat([torch.randint(0, 40, (n, 2), generator=g).float() / 40, torch.zeros(n, 1)], dim=1)
    clouds = [shell, clusters, dup, line, sheet]
    sizes = [n] * len(clouds)
    pos = torch.cat(clouds).float().contiguous()
    ptr = np.arange(0, (len(clouds) + 1) * n, n, dtype=np.int64)
    ref, optr = O.fps(pos.numpy(), ptr, 0.5)
    seg = Segments(sizes, DEV)
    cseg = Segments([ops.fps_count(n, 0.5)] * len(clouds), DEV)
    gap = torch.empty(len(clouds), dtype=torch.float32, device=DEV)
    idx = ops.fps(pos.to(DEV), seg.ptr, cseg.ptr, n, cseg.total, gap_out=gap)
    got = idx.cpu().numpy().astype(np.int64)
    for b in range(len(clouds)):
        assert np.array_equal(got[optr[b]:optr[b + 1]], ref[optr[b]:optr[b + 1]]), ("cloud", b)
    gp = gap.cpu().numpy()
    assert gp[0] > 0 and gp[1] > 0 and gp[2] == 0 and gp[3] > 0         # the smallest running maximum: what gn_fps_nested's shortcut rests on


def test_fps_nested_sample_is_the_plain_sample():
    """gn_fps_nested: the second level of the SA cascade (fps over the points the first level selected, in selection order) is a prefix wherever the
    first level's running maximum stayed positive -- and is sampled step by step where it did not (an example made of duplicated lattice points)"""
    g = torch.Generator().manual_seed(9)
    sizes = [6000, 2000, 777, 3000]
    clouds = [torch.rand(n, 3, generator=g) - 0.5 for n in sizes]
    clouds[1] = torch.randint(0, 4, (sizes[1], 3), generator=g).float() * 0.25          # 64 distinct points: the running maximum reaches 0
    pos = torch.cat(clouds).to(DEV)
    seg = Segments(sizes, DEV)
    m1 = [ops.fps_count(n, 0.5) for n in sizes]
    seg1 = Segments(m1, DEV)
    gap1 = torch.empty(len(sizes), dtype=torch.float32, device=DEV)
    idx1 = ops.fps(pos, seg.ptr, seg1.ptr, max(sizes), seg1.total, gap_out=gap1)
    assert torch.equal(idx1, ops.fps(pos, seg.ptr, seg1.ptr, max(sizes), seg1.total))
    g1 = gap1.cpu().numpy()
    assert g1[0] > 0 and g1[2] > 0 and g1[3] > 0 and g1[1] == 0
    pos1 = pos[idx1.long()]
    m2 = [ops.fps_count(n, 0.25) for n in m1]
    seg2 = Segments(m2, DEV)
    gap2 = torch.empty(len(sizes), dtype=torch.float32, device=DEV)
    plain = ops.fps(pos1, seg1.ptr, seg2.ptr, max(m1), seg2.total)
    nested = ops.fps(pos1, seg1.ptr, seg2.ptr, max(m1), seg2.total, gap_out=gap2, nested_gap=gap1)
    assert torch.equal(plain, nested)
    ref, _ = O.fps(pos1.cpu().numpy(), np.asarray(seg1.ptr.cpu().numpy(), dtype=np.int64), 0.25)
    assert np.array_equal(nested.cpu().numpy().astype(np.int64), ref)
    p1, p2 = seg1.ptr.cpu().numpy(), seg2.ptr.cpu().numpy()
    for b in (0, 2, 3):                                                                  # the prefix, as claimed
        assert np.array_equal(nested.cpu().numpy()[p2[b]:p2[b + 1]], p1[b] + np.arange(m2[b]))
    assert np.array_equal(gap2.cpu().numpy()[[0, 2, 3]], g1[[0, 2, 3]]) and gap2.cpu().numpy()[1] == 0
    # (the module cascade -- SAModule hands the first level's gap to the second through the position tensor it returns -- is held to the oracle's
    #  indices of BOTH levels by test_sa_module_graph_bit_exact and by every pipeline golden)


def test_fps_start_index():
    """random_start plumbing: any start index gives a valid greedy max-min sequence beginning at that point."""
    _, pos, batch = _ragged_cloud([500, 300], 21)
    seg, cseg = Segments([500, 300], DEV), Segments([250, 150], DEV)
    start = torch.tensor([17, 299], dtype=torch.int32, device=DEV)
    idx = ops.fps(pos.to(DEV), seg.ptr, cseg.ptr, 500, 400, start).cpu().numpy()
    assert idx[0] == 17 and idx[250] == 500 + 299
    for lo, hi, off, n in ((0, 250, 0, 500), (250, 400, 500, 300)):
        sel = idx[lo:hi] - off
        assert len(set(sel.tolist())) == len(sel) and sel.min() >= 0 and sel.max() < n
        p = pos[off:off + n].double()
        d = torch.cdist(p, p[torch.from_numpy(sel.astype(np.int64))])
        for k in (1, 5, 60):
            dmin = d[:, :k].min(dim=1)[0]
            assert abs(float(dmin.max()) - float(dmin[sel[k]])) < 1e-7


def test_segment_ptr():
    batch = torch.tensor([0, 0, 2, 2, 2, 5], dtype=torch.int64)
    ptr = ops.segment_ptr(batch.to(DEV), 7).cpu().numpy()
    assert list(ptr) == [0, 2, 2, 5, 5, 5, 6, 6]


@pytest.mark.parametrize("sizes,r,K", [([6000], 0.05, 64), ([3000, 2000], 0.1, 64), ([500, 3, 200], 0.1, 16)])
def test_ball_query_bit_exact(sizes, r, K):
    _, pos, batch = _ragged_cloud(sizes, 5)
    ptr = O.batch_to_ptr(batch.numpy())
    cidx, cptr = O.fps(pos.numpy(), ptr, 0.5)
    ref_nbr, ref_cnt = O.ball_query(pos.numpy(), ptr, cidx, cptr, r, K)
    seg, cseg = Segments(sizes, DEV), Segments(list(np.diff(cptr)), DEV)
    nbr, cnt = ops.ball_query(pos.to(DEV), seg.ptr, torch.from_numpy(cidx.astype(np.int32)).to(DEV), cseg.ptr, r, K)
    assert np.array_equal(cnt.cpu().numpy(), ref_cnt)
    assert np.array_equal(nbr.cpu().numpy(), ref_nbr)
    if r > 0.05:
        assert ref_cnt.max() == K        # the truncation rule is exercised


@pytest.mark.parametrize("k", [1, 3])
def test_knn_interpolate(k):
    sizes = [1500, 700]
    _, pos, batch = _ragged_cloud(sizes, 7)
    ptr = O.batch_to_ptr(batch.numpy())
    sidx, sptr = O.fps(pos.numpy(), ptr, 0.25)
    xs = torch.randn(len(sidx), 70, generator=torch.Generator().manual_seed(1))
    ps = pos[torch.from_numpy(sidx)]
    ref = O.knn_interpolate(xs.numpy(), ps.numpy(), sptr, pos.numpy(), ptr, k)
    out = ops.knn_interpolate(xs.to(DEV), ps.contiguous().to(DEV), Segments(list(np.diff(sptr)), DEV).ptr, pos.to(DEV),
                              Segments(sizes, DEV).ptr, k)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,K,N,relu,bn", [(1000, 6, 64, True, True), (777, 131, 128, True, True), (300, 259, 256, True, True),
                                            (513, 137, 137, True, True), (4096, 256, 1, True, True), (200, 256, 3, True, True),
                                            (129, 1280, 256, True, True), (64, 128, 192, False, False), (5, 1024, 1024, False, False),
                                            (100000, 32, 128, False, False)])
def test_linear_against_torch(M, K, N, relu, bn):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    sc = torch.rand(N, generator=g) + 0.5 if bn else None
    sh = torch.randn(N, generator=g) if bn else None
    ref = F.linear(x, w, b)
    if relu:
        ref = F.relu(ref)
    if bn:
        ref = ref * sc + sh
    xp = ops.new_rows(M, K, DEV)
    xp.copy_(x)
    wp = torch.zeros(N, ops.pad4(K), device=DEV)
    wp[:, :K] = w.to(DEV)
    out = ops.linear(xp, wp, b.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), relu=relu, K=K)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    # unaligned leading dimensions take the scalar loader
    out2 = ops.linear(x.to(DEV), w.to(DEV).contiguous(), b.to(DEV), None, None, relu=False)
    np.testing.assert_allclose(out2.cpu().numpy(), F.linear(x, w, b).numpy(), rtol=1e-5, atol=2e-5)


def test_nocs_head():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5000, 192, generator=g) * 3
    logits[7, 3 * 5 + 1] = logits[7, 3 * 9 + 1] = 50.0    # tie -> first maximum
    idx, conf, nocs = ops.nocs_head(logits.to(DEV), 64)
    ridx, rconf, rnocs = P.nocs_postprocess(logits, 64)
    assert np.array_equal(idx.cpu().numpy(), ridx.numpy()) and int(idx[7, 1]) == 5
    np.testing.assert_allclose(conf.cpu().numpy(), rconf.numpy(), rtol=1e-5, atol=1e-6)
    assert np.array_equal(nocs.cpu().numpy(), rnocs.numpy())


# ------------------------------------------------------------------------------------------------ PointNet++ / pipeline vs goldens
@pytest.mark.parametrize("name", ["small_max", "small_mean", "dress_g32"])
def test_pipeline_against_reference_goldens(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"ref_{name}.npz"))
    B, n, G, Q, seed, stride = [int(v) for v in g["meta"]]
    hp = S.default_hparams(grid=G, reduce_method=str(g["reduce_method"]))
    model = _model(hp, seed)
    x, pos, batch = S.synthetic_cloud(B, n, seed)
    data = Batch(sizes=[n] * B, x=x, pos=pos, batch=batch).to(DEV)
    p2 = model.pointnet2_forward(data)
    nd = p2["nocs_data"]
    # integer decisions: NOCS bin arg-max (exact up to fp32 near-ties of the logits, which we require to be absent here)
    bins, _, _ = ops.nocs_head(p2["per_point_logits"], 64)
    assert np.array_equal(bins.cpu().numpy().astype(np.int8), g["nocs_bin_idx"])
    np.testing.assert_allclose(p2["per_point_features"].cpu().numpy()[::stride], g["per_point_features"], rtol=0, atol=TOL)
    np.testing.assert_allclose(p2["per_point_logits"].cpu().numpy()[::stride], g["per_point_logits"], rtol=0, atol=TOL)
    np.testing.assert_allclose(nd.pred_confidence.cpu().numpy()[::stride], g["pred_confidence"], rtol=0, atol=TOL)
    assert np.array_equal(nd.pos.cpu().numpy()[::stride], g["pred_nocs"])
    np.testing.assert_allclose(p2["global_logits"].cpu().numpy(), g["global_logits"], rtol=0, atol=TOL)
    np.testing.assert_allclose(p2["global_feature"].cpu().numpy(), g["global_feature"], rtol=0, atol=TOL)
    vin = model.volume_agg(nd)
    u3 = model.unet3d_forward(p2)
    vol = u3["out_feature_volume"]
    assert vol.shape == (B, 128, G, G, G)
    if "in_feature_volume" in g:
        np.testing.assert_allclose(vin.cpu().numpy(), g["in_feature_volume"], rtol=0, atol=TOL)
        np.testing.assert_allclose(vol.cpu().numpy(), g["out_feature_volume"], rtol=0, atol=TOL)
    else:
        np.testing.assert_allclose(vin.cpu().numpy()[:, ::16, ::3, ::3, ::3], g["in_volume_probe"], rtol=0, atol=TOL)
        np.testing.assert_allclose(vol.cpu().numpy()[:, ::16, ::3, ::3, ::3], g["out_volume_probe"], rtol=0, atol=TOL)
    wnf = model.volume_lattice_forward(u3, Q)["pred_volume"]
    np.testing.assert_allclose(wnf[0].cpu().numpy(), g["wnf_volume"], rtol=0, atol=TOL)
    sq = torch.from_numpy(g["surf_query"]).to(DEV)
    np.testing.assert_allclose(model.surface_decoder_forward(u3, sq)["out_features"].cpu().numpy(), g["surf_out"], rtol=0, atol=TOL)
    np.testing.assert_allclose(model.volume_decoder_forward(u3, sq)["pred_volume_value"].cpu().numpy(), g["volq_out"], rtol=0, atol=TOL)
    if B > 1:       # round 6: a batch's queries go through three launches (one row set per blockIdx.y) -- bit for bit what the garment loop gives
        q3 = torch.rand(B, 1500, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
        for dec_name, fwd, key in (("surface_decoder", model.surface_decoder_forward, "out_features"), ("volume_decoder", model.volume_decoder_forward, "pred_volume_value")):
            dec = getattr(model, dec_name)
            batched = fwd(u3, q3)[key]
            dec.BATCH_ROWS_BYTES = 0                  # (instance attribute: forces the loop)
            try:
                looped = fwd(u3, q3)[key]
            finally:
                del dec.BATCH_ROWS_BYTES
            assert torch.equal(batched, looped), dec_name
    # the reference's own chunked query loop (predict.py:145-157) through the API-compatible path gives the same volume
    from garmentnets_amd.components.gridding import ArraySlicer, VirtualGrid
    gp = VirtualGrid(grid_shape=(Q,) * 3).get_grid_points(include_batch=False)
    out = torch.zeros(gp.shape[:-1], device=DEV)
    u3_0 = u3.select(0, 1)
    for sl in ArraySlicer(gp.shape, (64, 64, 64)):
        q = gp[tuple(sl)]
        out[tuple(sl)] = model.volume_decoder_forward(u3_0, q.to(DEV).view(1, -1, 3))["pred_volume_value"].view(*q.shape[:-1])
    assert torch.equal(out, wnf[0])
    # a plain dict holding the materialised 128-channel volume (what reference-side code may build) goes through the decoder's
    # literal order of operations (sample 128 channels, unfolded first layer): the same numbers within the budget
    lit = model.volume_decoder_forward({"out_feature_volume": vol[0:1]}, sq[0:1])["pred_volume_value"]
    np.testing.assert_allclose(lit.cpu().numpy(), g["volq_out"][0:1], rtol=0, atol=TOL)
    np.testing.assert_allclose(lit.cpu().numpy(), model.volume_decoder_forward(u3_0, sq[0:1])["pred_volume_value"].cpu().numpy(), rtol=0, atol=2e-5)


def test_hip_graph_replay_is_bit_identical():
    """graphs.GraphedDenseStages: the dense stages captured into a HIP graph (every C-ABI kernel launches on torch's current stream)
    and replayed on a different cloud == the eager path, bit for bit"""
    from garmentnets_amd.graphs import GraphedDenseStages
    hp = S.default_hparams(grid=16, reduce_method="max")
    model = _model(hp, 0)
    def mk(seed):
        x, pos, b = S.synthetic_cloud(2, 900, seed=seed)
        return Batch(sizes=[900, 900], x=x, pos=pos, batch=b).to(DEV)
    d0, d1 = mk(0), mk(1)
    with torch.no_grad():
        p2 = model.pointnet2_forward(d1)
        u3 = model.unet3d_forward(p2)
        ref = model.volume_lattice_forward(u3, 24)["pred_volume"].clone()
        ref_logits = p2["per_point_logits"].clone()
    g = GraphedDenseStages(model, d0, 24)
    gp2, gu3, gwnf = g(d1)
    assert torch.equal(gwnf, ref) and torch.equal(gp2["per_point_logits"], ref_logits)
    gp2, gu3, gwnf = g(d0)                           # replay again with the capture-time cloud: different result, same buffers
    assert not torch.equal(gwnf, ref)
    with pytest.raises(ValueError):
        g(Batch(sizes=[900], x=d0.x[:900], pos=d0.pos[:900], batch=d0.batch[:900]))


def test_bf16x2_preview_mode_warns_and_is_tracked_against_the_contract(golden_dir):
    """the PREVIEW arithmetic announces itself (RuntimeWarning at selection), and its distance to the 1e-4 contract stays visible: an
    xfail (non-strict) that turns into an XPASS the day the mode meets the tolerance"""
    with pytest.warns(RuntimeWarning, match="PREVIEW"):
        ar = AR.Arith.named("bf16x2")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        AR.Arith.named("f16x2"), AR.Arith.named("bf16x3"), AR.Arith.named("fp32")            # the contract-grade modes are silent
    g = np.load(os.path.join(golden_dir, "ref_dress_g32.npz"))
    B, n, G, Q, seed, stride = [int(v) for v in g["meta"]]
    model = _model(S.default_hparams(grid=G, reduce_method=str(g["reduce_method"])), seed)
    x, pos, batch = S.synthetic_cloud(B, n, seed)
    model.arith = ar
    u3 = model.unet3d_forward(model.pointnet2_forward(Batch(sizes=[n] * B, x=x, pos=pos, batch=batch).to(DEV)))
    err = np.abs(model.volume_lattice_forward(u3, Q)["pred_volume"][0].cpu().numpy() - g["wnf_volume"]).max()
    if err > TOL:
        pytest.xfail(f"bf16x2 preview arithmetic: WNF error {err:.2e} > {TOL:.0e} (known; the mode is not a default anywhere)")


@pytest.mark.filterwarnings("ignore:garmentnets_amd. conv_mode bf16x2:RuntimeWarning")
@pytest.mark.parametrize("planes", [0, 4, 3, 2])
def test_pipeline_conv_modes(golden_dir, planes):
    """every conv arithmetic (0 = fp32 MFMA, 4 = f16x2 default, 3 / 2 = bf16 planes): the whole pipeline against the reference
    goldens, same 1e-4 budget on the WNF"""
    g = np.load(os.path.join(golden_dir, "ref_dress_g32.npz"))
    B, n, G, Q, seed, stride = [int(v) for v in g["meta"]]
    model = _model(S.default_hparams(grid=G, reduce_method=str(g["reduce_method"])), seed)
    x, pos, batch = S.synthetic_cloud(B, n, seed)
    data = Batch(sizes=[n] * B, x=x, pos=pos, batch=batch).to(DEV)
    model.arith = model.arith.replace(conv_mode=planes)          # per-model arithmetic: nothing global is switched
    p2 = model.pointnet2_forward(data)
    u3 = model.unet3d_forward(p2)
    wnf = model.volume_lattice_forward(u3, Q)["pred_volume"][0].cpu().numpy()
    err_vol = np.abs(u3["out_feature_volume"].cpu().numpy()[:, ::16, ::3, ::3, ::3] - g["out_volume_probe"]).max()
    err_wnf = np.abs(wnf - g["wnf_volume"]).max()
    print(f"conv mode={planes}: feature-volume err {err_vol:.2e}, WNF err {err_wnf:.2e}")
    # bf16x2 is a labelled PREVIEW arithmetic (arith.py, bench.py --conv-mode help): it is not held to the 1e-4 contract
    assert err_wnf <= (TOL if planes != 2 else 3 * TOL) and err_vol <= (TOL if planes != 2 else 5 * TOL)


def test_pipeline_ragged_batch_against_oracle():
    """garments of different sizes in one batch (sorted batch vector, no host-side sizes given): every stage vs the oracle"""
    hp = S.default_hparams(grid=16, reduce_method="mean")
    sd = S.synthetic_state_dict(hp, 7)
    sizes = [2500, 700, 1300]
    x, pos, batch = _ragged_cloud(sizes, 31)
    with torch.no_grad():
        ref = P.pointnet2_forward(sd, hp, x, pos, batch)
        rvol = P.unet3d(sd, hp["unet3d_params"], P.volume_agg(sd, hp["volume_agg_params"], ref["nocs_data"], 3))
    model = _model(hp, 7)
    p2 = model.pointnet2_forward(Batch(x=x, pos=pos, batch=batch).to(DEV))          # sizes derived from the batch vector
    assert p2["nocs_data"].sizes == sizes
    bins, _, _ = ops.nocs_head(p2["per_point_logits"], 64)
    assert np.array_equal(bins.cpu().numpy(), ref["nocs_data"]["nocs_bin_idx"].numpy())
    np.testing.assert_allclose(p2["per_point_features"].cpu().numpy(), ref["per_point_features"].numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(p2["global_feature"].cpu().numpy(), ref["global_feature"].numpy(), rtol=0, atol=TOL)
    vol = model.unet3d_forward(p2)["out_feature_volume"]
    np.testing.assert_allclose(vol.cpu().numpy(), rvol.numpy(), rtol=0, atol=TOL)


def test_batched_isosurface_tail_equals_per_garment():
    """wnf_batch_to_meshes_gpu (one set of launches and one host synchronisation for the batch) == wnf_to_mesh_gpu per garment, bit for
    bit, including the error contract (level outside a volume's range -> ValueError entry); an odd lattice (Q^3 % 4 != 0: the volumes
    of the batch start at addresses that are not 16-byte aligned) as well"""
    from garmentnets_amd.common import marching_cubes_util as MCU
    for Q in (24, 23):
        base = torch.from_numpy(S.shell_volume(Q)).float()
        noise = torch.rand(3, Q, Q, Q, generator=torch.Generator().manual_seed(4)) * 0.05
        vols = (base[None] + noise).to(DEV)
        vols[1] = vols[1] * 0.2                          # range [0, ~0.21]: level 0.5 is outside
        batch = MCU.wnf_batch_to_meshes_gpu(vols, 0.5, 0.5, "ascent")
        assert isinstance(batch[1], ValueError)
        for b in (0, 2):
            one = MCU.wnf_to_mesh_gpu(vols[b], 0.5, 0.5, "ascent")
            for k in one:
                assert torch.equal(one[k], batch[b][k]), (Q, k)
        desc = MCU.wnf_batch_to_meshes_gpu(vols, 0.5, 0.5, "descent")
        assert torch.equal(desc[0]["faces"], torch.flip(batch[0]["faces"], dims=[1]))


@pytest.mark.parametrize("Q,B", [(32, 5), (20, 3), (128, 2)])
def test_batched_iso_operators_equal_single_volume_calls(Q, B):
    """gn_ggm3d_batch / gn_minmax_batch / gn_mc33_batch (a volume per blockIdx.y) against the one-volume entry points, bit for bit:
    shells of different radii (different vertex / face counts per volume), one volume without any surface, one noisy volume"""
    g = torch.Generator().manual_seed(Q + B)
    base = torch.from_numpy(S.shell_volume(Q)).float()
    vols = torch.stack([base * (0.6 + 0.2 * b) + 0.02 * b * torch.rand(Q, Q, Q, generator=g) for b in range(B)])
    vols[1] = 0.1                                                       # constant: no cell straddles the level
    vols = vols.to(DEV)
    cap_v, cap_f = 6 * Q * Q, 12 * Q * Q + 64
    ggm = ops.ggm3d_batch(vols, 0.5)
    mm = ops.minmax_batch(vols)
    verts, faces, normals, values, counts = ops.mc33_batch(vols, 0.5, cap_v, cap_f)
    cnt = counts.cpu().numpy()
    assert cnt[1].tolist() == [0, 0] and len({int(c) for c in cnt[:, 0]}) >= 2
    for b in range(B):
        assert torch.equal(ggm[b], ops.ggm3d(vols[b].contiguous(), 0.5))
        assert torch.equal(mm[b], ops.minmax(vols[b].contiguous()))
        v1, f1, n1, a1, c1 = ops.mc33(vols[b].contiguous(), 0.5, cap_v, cap_f)
        assert torch.equal(counts[b], c1)
        nv, nf = int(cnt[b, 0]), int(cnt[b, 1])
        assert nv <= cap_v and nf <= cap_f
        assert torch.equal(verts[b, :nv], v1[:nv]) and torch.equal(faces[b, :nf], f1[:nf])
        assert torch.equal(normals[b, :nv], n1[:nv]) and torch.equal(values[b, :nv], a1[:nv])
    assert ops.mc33_batch(vols[:0], 0.5, 8, 8)[4].shape == (0, 2)


@pytest.mark.parametrize("Q,B", [(32, 5), (20, 3), (128, 3)])
def test_ggm_range_rides_along(Q, B):
    """gn_ggm3d_batch_ex (round 6): the gradient magnitude is gn_ggm3d_batch's bit for bit, the (min, max) record that rides on the fused launch's
    staging pass is gn_minmax_batch's and numpy's; a NaN anywhere in a volume gives NaN in ITS record only (numpy.min / numpy.max: what skimage's
    level check evaluates, and what predict reads its NaN flag from); infinities are ordinary extremes"""
    g = torch.Generator().manual_seed(7 * Q + B)
    base = torch.from_numpy(S.shell_volume(Q)).float()
    vols = torch.stack([base * (0.6 + 0.2 * b) - 0.3 * b + 0.02 * b * torch.rand(Q, Q, Q, generator=g) for b in range(B)])
    vols[1] = -0.0
    vols = vols.to(DEV)
    ggm, rng = ops.ggm3d_batch_range(vols, 0.5)
    assert torch.equal(ggm, ops.ggm3d_batch(vols, 0.5))
    assert torch.equal(rng, ops.minmax_batch(vols))
    host = vols.cpu().numpy()
    assert np.array_equal(rng.cpu().numpy(), np.stack([host.reshape(B, -1).min(1), host.reshape(B, -1).max(1)], 1))
    assert rng[1].tolist() == [0.0, 0.0]
    # a NaN in the last voxel of volume 0 (a corner: seen through the halo replication of one tile only), infinities in volume 2
    bad = vols.clone()
    bad[0, -1, -1, -1] = float("nan")
    bad[2, 3, 4, 5] = float("inf")
    bad[2, 5, 4, 3] = float("-inf")
    _, r2 = ops.ggm3d_batch_range(bad, 0.5)
    r2 = r2.cpu().numpy()
    assert np.isnan(r2[0]).all() and not np.isnan(r2[1:]).any()
    assert r2[2].tolist() == [float("-inf"), float("inf")] and np.array_equal(r2[1], rng[1].cpu().numpy())
    mm = ops.minmax_batch(bad).cpu().numpy()
    assert np.isnan(mm[0]).all() and mm[2].tolist() == [float("-inf"), float("inf")]
    job = MCU.IsoBatchJob(Q)
    job.enqueue(bad)
    assert bool(job.any_nan())
    job = MCU.IsoBatchJob(Q)
    job.enqueue(vols)
    assert not bool(job.any_nan())


@pytest.mark.parametrize("Q", [32, 128])
def test_ggm_fp32_accumulation_is_held_to_a_tolerance(Q):
    """Arith.ggm_fp32 / gn_ggm3d_batch_ex(accum_bits=32): the taps accumulate in fp32 in scipy's operation order -- no bit parity with scipy, held to
    2e-6 of the volume's largest gradient magnitude (measured 3e-7 .. 5e-7: three passes of 5 taps at 6e-8 each); the range record is the same"""
    g = torch.Generator().manual_seed(Q)
    base = torch.from_numpy(S.shell_volume(Q)).float()
    vols = torch.stack([base, base * 3.0 + 0.1 * torch.rand(Q, Q, Q, generator=g), torch.randn(Q, Q, Q, generator=g)]).to(DEV)
    g64, r64 = ops.ggm3d_batch_range(vols, 0.5, 64)
    g32, r32 = ops.ggm3d_batch_range(vols, 0.5, 32)
    assert torch.equal(r64, r32)
    for b in range(vols.shape[0]):
        top = float(g64[b].max())
        err = float((g32[b].double() - g64[b].double()).abs().max())
        assert 0 < err <= 2e-6 * top, (b, err, top)
    with pytest.raises(ValueError):
        ops.ggm3d_batch_range(vols, 0.5, 16)
    with pytest.raises(ValueError):                  # the 8-pass form (kernel radius above 2) has no fp32 variant
        ops.ggm3d_batch_range(vols, 1.0, 32)
    g1, r1 = ops.ggm3d_batch_range(vols, 1.0, 64)    # ... and takes its range from gn_minmax_batch
    assert torch.equal(g1, ops.ggm3d_batch(vols, 1.0)) and torch.equal(r1, r64)


def test_degenerate_sizes():
    """zero-row launches are no-ops; k-NN with fewer sources than k uses what exists (as the oracle does)"""
    z = ops.linear(torch.zeros(0, 8, device=DEV), torch.zeros(4, 8, device=DEV))
    assert z.shape == (0, 4)
    assert ops.trilinear_sample(torch.zeros(2, 2, 2, 4, device=DEV), query=torch.zeros(0, 3, device=DEV)).shape == (0, 4)
    ps = torch.tensor([[0.0, 0, 0], [1.0, 0, 0]])
    xs = torch.tensor([[1.0, 2.0], [3.0, 5.0]])
    pq = torch.tensor([[0.25, 0, 0], [0.9, 0.1, 0], [0.0, 0.0, 0.0]])
    ref = O.knn_interpolate(xs.numpy(), ps.numpy(), np.array([0, 2]), pq.numpy(), np.array([0, 3]), 3)
    out = ops.knn_interpolate(xs.to(DEV), ps.to(DEV), Segments([2], DEV).ptr, pq.to(DEV), Segments([3], DEV).ptr, 3)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert np.allclose(out[2].cpu().numpy(), [1.0, 2.0], atol=1e-5)       # coincident point: weight 1/1e-16 dominates
    # the split-operand entries: empty batch / no rows / no queries are no-ops; bad modes and widths are refused with GN_EINVAL
    w = torch.randn(32, 16, 3, 3, 3)
    pk = ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV)
    e = ops.conv3d_gcr_split(torch.zeros(0, 4, 8, 8, 16, device=DEV), None, torch.zeros(0, 16, device=DEV), torch.zeros(0, 16, device=DEV), pk, 32)
    assert e.shape == (0, 4, 8, 8, 32)
    with pytest.raises(ValueError):
        ops.conv3d_gcr_split(torch.zeros(1, 4, 8, 8, 16, device=DEV), None, torch.ones(1, 16, device=DEV), torch.zeros(1, 16, device=DEV),
                             ops.SplitPack(pk.tensor, 7, pk.out_scale), 32)
    raw = [(torch.randn(256, 32), torch.randn(256), None, None), (torch.randn(256, 256), torch.randn(256), None, None), (torch.randn(1, 256), torch.randn(1), None, None)]
    dp = ops.pack_decode_split(raw).to(DEV)
    assert ops.implicit_decode_split(ops.new_rows(0, 32, DEV), dp).shape == (0, 1)
    with pytest.raises(ValueError):
        ops.implicit_decode_split(ops.new_rows(5, 64, DEV), dp)            # 64-wide rows: not a packed first-layer width
    i0, d0 = ops.nearest_neighbor(torch.zeros(0, 3, device=DEV), torch.zeros(4, 3, device=DEV))
    assert i0.shape == (0,) and d0.shape == (0,)
    with pytest.raises(ValueError):
        ops.nearest_neighbor(torch.zeros(3, 3, device=DEV), torch.zeros(0, 3, device=DEV))


def test_sa_module_graph_bit_exact():
    """fps order and ball-query tables of both set-abstraction levels, on the BASELINE cloud size."""
    hp = S.default_hparams()
    sd = S.synthetic_state_dict(hp, 0)
    x, pos, batch = S.synthetic_cloud(2, 6000, seed=11)
    ref = P.pointnet2_nocs_forward(sd, hp["pointnet2_params"], x, pos, batch, return_intermediates=True)["_inter"]
    model = _model(hp, 0)
    net = model.pointnet2_nocs
    net(Batch(sizes=[6000, 6000], x=x, pos=pos, batch=batch).to(DEV))
    for mod, key in ((net.sa1_module, "sa1"), (net.sa2_module, "sa2")):
        idx, nbr = mod.last_graph
        # sa2 indices are relative to the sa1 point set in both implementations
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref[key + "_idx"])
        assert np.array_equal(nbr.cpu().numpy(), ref[key + "_nbr"])


# ------------------------------------------------------------------------------------------------ gridding
@pytest.mark.parametrize("reduce,G", [("max", 32), ("mean", 32), ("mean", 128)])
def test_grid_scatter(reduce, G):
    g = torch.Generator().manual_seed(G)
    N, B, C = 5000, 2, 128
    bins = torch.randint(0, 64, (N, 3), generator=g)
    bins[:40] = bins[0]                      # heavy collisions
    nocs = bins.float() * (1.0 / 63.0)
    feat = torch.randn(N, C, generator=g)
    batch = torch.sort(torch.randint(0, B, (N,), generator=g))[0]
    gi = P.points_grid_idxs(nocs, [0, 0, 0], [1, 1, 1], (G,) * 3)
    flat_ref = ((batch * G + gi[:, 0]) * G + gi[:, 1]) * G + gi[:, 2]
    sim = torch.randn(N, 3, generator=g)
    conf = torch.rand(N, 3, generator=g)
    feats, flat = ops.grid_features(feat.to(DEV), nocs.to(DEV), sim.to(DEV), conf.to(DEV), batch.to(DEV), (0, 0, 0), (1, 1, 1), (G,) * 3)
    assert np.array_equal(flat.cpu().numpy().astype(np.int64), flat_ref.numpy())          # cell indices: exact
    ref_feats = torch.cat([feat, nocs - P.idxs_to_points(gi, [0, 0, 0], [1, 1, 1], (G,) * 3), sim, conf], dim=1)
    assert torch.equal(feats.cpu(), ref_feats)
    src = feats[:, :C].contiguous()
    vol, (ssum, ssq, cps) = ops.grid_scatter(src, flat, B, (G,) * 3, reduce, with_stats=True)
    # statistics from the occupied cells only == statistics of the whole (mostly empty) volume
    full = vol.reshape(B, -1, C).double()
    assert cps == G ** 3
    np.testing.assert_allclose(ssum.cpu().numpy(), full.sum(1).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(ssq.cpu().numpy(), (full ** 2).sum(1).cpu().numpy(), rtol=1e-6, atol=1e-4)
    red = {"max": "amax", "mean": "mean"}[reduce]
    ref = torch.zeros(B * G ** 3, C).scatter_reduce(0, flat_ref.unsqueeze(1).expand(-1, C), feat, red, include_self=False)
    got = vol.reshape(-1, C).cpu()
    if reduce == "max":
        assert torch.equal(got, ref)                                    # order-independent -> bit exact
    else:
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    assert int((got.abs().sum(1) > 0).sum()) == len(torch.unique(flat_ref))                 # empty cells stay 0


# ------------------------------------------------------------------------------------------------ UNet pieces
@pytest.mark.parametrize("C0,C1,Cout,dims", [(128, 0, 128, (8, 8, 8)), (32, 0, 64, (4, 12, 20)), (64, 128, 64, (8, 8, 16)),
                                              (16, 0, 32, (1, 1, 1)), (128, 256, 128, (2, 2, 2)), (32, 0, 256, (5, 9, 3))])
def test_conv3d_gcr_against_torch(C0, C1, Cout, dims):
    g = torch.Generator().manual_seed(C0 + C1 + Cout)
    B, (D, H, W) = 2, dims
    x0 = torch.randn(B, C0, D, H, W, generator=g)
    x1 = torch.randn(B, C1, D // 2, H // 2, W // 2, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) / (27 * (C0 + C1)) ** 0.5
    gamma = torch.rand(C0 + C1, generator=g) + 0.5
    beta = torch.randn(C0 + C1, generator=g)
    xin = x0 if x1 is None else torch.cat((x0, F.interpolate(x1, size=(D, H, W), mode="nearest")), dim=1)
    ref = F.relu(F.conv3d(F.group_norm(xin, 8, gamma, beta, eps=1e-5), w, None, padding=1))
    s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    s1 = None if x1 is None else x1.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    st0 = ops.channel_stats(s0)
    st1 = None if s1 is None else ops.channel_stats(s1)
    a, d = ops.groupnorm_affine(st0, st1, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    wp = ops.pack_conv_weight(w).to(DEV)
    out, (osum, osq, V) = ops.conv3d_gcr(s0, s1, a, d, wp, Cout, relu=True, with_stats=True)
    np.testing.assert_allclose(out.permute(0, 4, 1, 2, 3).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    # the epilogue's GroupNorm statistics of the output == a separate statistics pass over it
    rs, rq, rV = ops.channel_stats(out)
    assert V == rV == D * H * W
    np.testing.assert_allclose(osum.cpu().numpy(), rs.cpu().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(osq.cpu().numpy(), rq.cpu().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(osum.cpu().numpy(), ref.double().sum(dim=(2, 3, 4)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("planes,tol", [(3, 2e-5), (4, 2e-5), (2, 2e-4)])
@pytest.mark.parametrize("C0,C1,Cout,dims", [(128, 0, 128, (8, 8, 8)), (64, 128, 64, (8, 8, 16)), (16, 0, 32, (3, 5, 9)), (32, 0, 256, (4, 8, 8))])
def test_conv3d_split_against_torch(C0, C1, Cout, dims, planes, tol):
    """opt-in split-precision conv (bf16 planes on the matrix cores) against torch fp32 and against the fp32-MFMA kernel"""
    g = torch.Generator().manual_seed(C0 + C1 + Cout + planes)
    B, (D, H, W) = 2, dims
    x0 = torch.randn(B, C0, D, H, W, generator=g)
    x1 = torch.randn(B, C1, D // 2, H // 2, W // 2, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) / (27 * (C0 + C1)) ** 0.5
    gamma = torch.rand(C0 + C1, generator=g) + 0.5
    beta = torch.randn(C0 + C1, generator=g)
    xin = x0 if x1 is None else torch.cat((x0, F.interpolate(x1, size=(D, H, W), mode="nearest")), dim=1)
    ref64 = F.relu(F.conv3d(F.group_norm(xin.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    s1 = None if x1 is None else x1.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    a, d = ops.groupnorm_affine(ops.channel_stats(s0), None if s1 is None else ops.channel_stats(s1), 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    out32 = ops.conv3d_gcr(s0, s1, a, d, ops.pack_conv_weight(w).to(DEV), Cout).permute(0, 4, 1, 2, 3).cpu().double()
    wps = ops.pack_conv_weight_split(w, planes).to(DEV)
    out, (osum, osq, V) = ops.conv3d_gcr_split(s0, s1, a, d, wps, Cout, with_stats=True)
    got = out.permute(0, 4, 1, 2, 3).cpu().double()
    e_split, e_f32 = (got - ref64).abs().max().item(), (out32 - ref64).abs().max().item()
    assert e_split <= tol, (e_split, e_f32)
    print(f"mode {planes}: err vs fp64 {e_split:.2e} (fp32-MFMA kernel {e_f32:.2e})")
    if planes != 2:
        assert e_split <= 2 * max(e_f32, 2e-6), (e_split, e_f32)      # as accurate as the fp32 matrix-core kernel
    np.testing.assert_allclose(osum.cpu().numpy(), got.sum(dim=(2, 3, 4)).numpy(), rtol=1e-5, atol=1e-3)


def test_conv3d_f16x2_range_contract():
    """fp16 planes.  (1) The raw kernel without the sample's range normalisation: GroupNorm outputs beyond +-65504 surface as inf/NaN
    (never as a wrong finite number).  (2) The product path (groupnorm_affine(with_act_scale=True) + the epilogue's exact undo): any
    affine gain, 1e-6 ... 1e6, gives fp32-class results -- overflow is impossible by construction and tiny activations keep both planes
    normal.  bf16x3 has fp32's range either way."""
    g = torch.Generator().manual_seed(9)
    B, C0, Cout, (D, H, W) = 1, 32, 32, (4, 8, 8)
    x0 = torch.randn(B, C0, D, H, W, generator=g)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    st = ops.channel_stats(s0)
    pk, pk3 = ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), ops.pack_conv_weight_split(w, ops.SPLIT_BF16X3).to(DEV)
    for gain in (1.0e-6, 1.0e-3, 1.0, 1.0e4, 1.0e6):
        gamma, beta = torch.full((C0,), gain), torch.zeros(C0)
        ref64 = F.relu(F.conv3d(F.group_norm(x0.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
        scale = ref64.abs().max().item()
        a, d = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
        raw = ops.conv3d_gcr_split(s0, None, a, d, pk, Cout).permute(0, 4, 1, 2, 3).cpu().double()
        out3 = ops.conv3d_gcr_split(s0, None, a, d, pk3, Cout).permute(0, 4, 1, 2, 3).cpu().double()
        assert torch.isfinite(out3).all() and (out3 - ref64).abs().max().item() <= 1e-5 * scale
        if gain >= 1.0e6:
            assert not torch.isfinite(raw).all()
        a2, d2, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV), with_act_scale=True)
        m = float(torch.log2(inv[0]))
        assert m == round(m)                                                     # an exact power of two
        out = ops.conv3d_gcr_split(s0, None, a2, d2, pk, Cout, act_inv=inv).permute(0, 4, 1, 2, 3).cpu().double()
        err = (out - ref64).abs().max().item()
        print(f"f16x2 gain {gain:g}: act scale 2^{-m:.0f}, err/scale {err / scale:.2e}")
        assert torch.isfinite(out).all() and err <= 2e-6 * scale


def test_conv3d_f16x2_small_activations_and_weight_outliers():
    """the two cases the per-tensor scale of round 1 lost precision on.  (i) activations whose variance is dominated by GroupNorm's eps
    (inputs ~1e-4 ... 1e-6: the normalised values are ~1e-2 ... 1e-4, the second fp16 plane of an unscaled split is subnormal);
    (ii) a weight tensor with a x1000 outlier in one output channel and one with a x1000 outlier ROW (every other row would sit in the
    subnormal second plane under a per-tensor scale).  Error vs fp64 must stay fp32-class RELATIVE TO EACH OUTPUT CHANNEL's own scale."""
    g = torch.Generator().manual_seed(11)
    B, C0, Cout, (D, H, W) = 2, 32, 64, (4, 8, 8)
    gamma, beta = 1.0 + 0.1 * torch.randn(C0, generator=g), torch.zeros(C0)
    w0 = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    cases = []
    for mag in (1e-3, 1e-4, 1e-6):
        cases.append((f"x~{mag:g}", torch.randn(B, C0, D, H, W, generator=g) * mag, w0))
    w1 = w0.clone(); w1[3, 5, 1, 1, 1] *= 1000.0
    w2 = w0.clone(); w2[7] *= 1000.0
    x1 = torch.randn(B, C0, D, H, W, generator=g)
    cases += [("one weight x1000", x1, w1), ("one row x1000", x1, w2)]
    for name, x0, w in cases:
        s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
        ref64 = F.relu(F.conv3d(F.group_norm(x0.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
        a, d, inv = ops.groupnorm_affine(ops.channel_stats(s0), None, 8, 1e-5, gamma.to(DEV), beta.to(DEV), with_act_scale=True)
        out = ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), Cout, act_inv=inv)
        out = out.permute(0, 4, 1, 2, 3).cpu().double()
        a0, d0 = ops.groupnorm_affine(ops.channel_stats(s0), None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
        o32 = ops.conv3d_gcr(s0, None, a0, d0, ops.pack_conv_weight(w).to(DEV), Cout).permute(0, 4, 1, 2, 3).cpu().double()
        ch_scale = ref64.abs().amax(dim=(0, 2, 3, 4)).clamp_min(1e-30)                         # per output channel
        e16 = ((out - ref64).abs().amax(dim=(0, 2, 3, 4)) / ch_scale).max().item()
        e32 = ((o32 - ref64).abs().amax(dim=(0, 2, 3, 4)) / ch_scale).max().item()
        print(f"{name}: per-channel relative err f16x2 {e16:.2e}, fp32-MFMA kernel {e32:.2e}")
        assert e16 <= max(2 * e32, 2e-6)


@pytest.mark.parametrize("planes", [4, 2])
@pytest.mark.parametrize("C0,C1,Cout,dims", [(32, 0, 128, (32, 32, 64)), (16, 32, 128, (30, 34, 62)), (16, 0, 256, (16, 32, 64)),
                                              (32, 0, 32, (64, 32, 64)), (16, 16, 32, (62, 30, 66))])
def test_conv3d_split_wide_variant_against_torch(C0, C1, Cout, dims, planes):
    """shapes large enough (>= 512 workgroups) to dispatch the big-volume variants -- conv3d_split_wide_kernel (Cout % 128 == 0: 8 waves,
    shared double-buffered halo, staging overlapped with the MFMA stream) and the tall-tile 32-wide kernel (Cout == 32, 8 x 8 x 8 tiles):
    against torch fp64, the fp32-MFMA kernel, ragged dims, the upsampled source"""
    g = torch.Generator().manual_seed(C0 + C1 + Cout + planes)
    B, (D, H, W) = 2, dims
    if Cout % 128 == 0:
        assert -(-D // 4) * -(-H // 8) * -(-W // 8) * (Cout // 128) * B >= 512       # conv3d_split_wide_kernel
    else:
        assert Cout == 32 and -(-D // 8) * -(-H // 8) * -(-W // 8) * B >= 512         # the tall-tile (8 x 8 x 8) variant of the 32-wide kernel
    x0 = torch.randn(B, C0, D, H, W, generator=g)
    x1 = torch.randn(B, C1, D // 2, H // 2, W // 2, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) / (27 * (C0 + C1)) ** 0.5
    gamma = torch.rand(C0 + C1, generator=g) + 0.5
    beta = torch.randn(C0 + C1, generator=g)
    xin = x0 if x1 is None else torch.cat((x0, F.interpolate(x1, size=(D, H, W), mode="nearest")), dim=1)
    ref64 = F.relu(F.conv3d(F.group_norm(xin.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    s1 = None if x1 is None else x1.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    a, d = ops.groupnorm_affine(ops.channel_stats(s0), None if s1 is None else ops.channel_stats(s1), 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    out32 = ops.conv3d_gcr(s0, s1, a, d, ops.pack_conv_weight(w).to(DEV), Cout).permute(0, 4, 1, 2, 3).cpu().double()
    out, (osum, osq, V) = ops.conv3d_gcr_split(s0, s1, a, d, ops.pack_conv_weight_split(w, planes).to(DEV), Cout, with_stats=True)
    got = out.permute(0, 4, 1, 2, 3).cpu().double()
    e_split, e_f32 = (got - ref64).abs().max().item(), (out32 - ref64).abs().max().item()
    print(f"wide mode {planes}: err vs fp64 {e_split:.2e} (fp32-MFMA kernel {e_f32:.2e})")
    assert e_split <= (2 * max(e_f32, 2e-6) if planes == 4 else 2e-4), (e_split, e_f32)
    np.testing.assert_allclose(osum.cpu().numpy(), got.sum(dim=(2, 3, 4)).numpy(), rtol=1e-5, atol=2e-2)
    rs, rq, rV = ops.channel_stats(out)
    np.testing.assert_allclose(osq.cpu().numpy(), rq.cpu().numpy(), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("C,dims", [(16, (6, 8, 10)), (32, (8, 8, 8)), (128, (4, 6, 2)), (20, (4, 4, 4))])
def test_maxpool(C, dims):
    x = torch.randn(2, C, *dims, generator=torch.Generator().manual_seed(C))
    ref = F.max_pool3d(x, 2)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    assert torch.equal(ops.maxpool3d_2(xc).permute(0, 4, 1, 2, 3).cpu(), ref)
    if 256 % (C // 4) == 0:
        out, (s, q, V) = ops.maxpool3d_2(xc, with_stats=True)
        assert torch.equal(out.permute(0, 4, 1, 2, 3).cpu(), ref) and V == ref[0, 0].numel()
        np.testing.assert_allclose(s.cpu().numpy(), ref.double().sum(dim=(2, 3, 4)).numpy(), rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(q.cpu().numpy(), (ref.double() ** 2).sum(dim=(2, 3, 4)).numpy(), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("name", ["unet_g8", "unet_g16"])
def test_unet_against_reference_module(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"ref_{name}.npz"))
    G, B, seed = [int(v) for v in g["meta"]]
    model = _model(S.default_hparams(grid=G), seed)
    x = torch.randn(B, 128, G, G, G, generator=torch.Generator().manual_seed(seed))
    y = model.unet_3d(x.to(DEV)).cpu().numpy()
    # dense N(0,1) input: outputs reach |y| ~ 5, so the 1e-4 budget is applied relative to magnitude as well; the
    # fp64 restatement shows both fp32 implementations sit within a few 1e-5 of the exact result
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=TOL)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in S.synthetic_state_dict(S.default_hparams(grid=G), seed).items()}
    with torch.no_grad():
        y64 = P.unet3d(sd64, S.default_hparams(grid=G)["unet3d_params"], x.double()).numpy()
    assert np.abs(y - y64).max() <= 2 * max(np.abs(g["y"] - y64).max(), 2.5e-5)   # as accurate as the reference's own fp32 path


@pytest.mark.parametrize("B,dims,C0,Cout,scattered", [(2, (8, 16, 16), 32, 128, False), (2, (8, 16, 16), 32, 128, True), (1, (4, 8, 8), 128, 128, False),
                                                      (2, (12, 8, 24), 64, 256, False), (1, (16, 16, 16), 128, 128, True), (1, (8, 8, 40), 256, 128, False)])
def test_conv3d_winograd_against_fp64(B, dims, C0, Cout, scattered):
    """Winograd F(2,3)-along-x form of the 128-wide conv (gn_conv3d_gcr_split_wino, csrc/unet_wino.hip; layer: components/unet3d.py:53-76): both
    operand forms -- literal (static transformed pack) and affine-in-weights (gn_conv_affine_pack_wino) -- against torch in fp64, next to the direct
    f16x2 form and the fp32-MFMA kernel.  The bar is the f16x2 contract: error <= 2x the fp32-MFMA kernel's.  Volumes of one or two tiles per axis:
    every voxel sits on a face somewhere (the border-class bias table and the zero padding of the transformed halo)."""
    g = torch.Generator().manual_seed(C0 + Cout + dims[2])
    D, H, W = dims
    x = torch.randn(B, C0, D, H, W, generator=g)
    if scattered:
        x = x * (torch.rand(B, 1, D, H, W, generator=g) < 0.05)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    gamma, beta = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    ref = F.relu(F.conv3d(F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    s0 = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    st = ops.channel_stats(s0)
    a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV), with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    cl = lambda t: t.permute(0, 4, 1, 2, 3).cpu().double()
    err = lambda t: float((cl(t) - ref).abs().max())
    e32 = err(ops.conv3d_gcr(s0, None, a0, d0, ops.pack_conv_weight(w).to(DEV), Cout))
    e_dir = err(ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), Cout, act_inv=inv))
    yw, (sm, sq, V) = ops.conv3d_gcr_split_wino(s0, a, d, ops.pack_conv_weight_split_wino(w).to(DEV), Cout, act_inv=inv, with_stats=True)
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino_kernel<true>"
    wd = w.to(DEV).contiguous()
    prep = ops.conv_affine_pack(wd, a0, d0, st, wino=True)
    yr, (smr, sqr, _) = ops.conv3d_gcr_split_persample(s0, prep, with_stats=True)
    e_w, e_r = err(yw), err(yr)
    print(f"B={B} {dims} {C0}->{Cout} {'scattered' if scattered else 'dense'}: err vs fp64: fp32-MFMA {e32:.2e}, direct f16x2 {e_dir:.2e}, Winograd literal {e_w:.2e}, "
          f"Winograd affine-in-weights {e_r:.2e}")
    bar = 2 * max(e32, 2e-6)
    assert e_w <= bar and e_r <= bar
    for y, s_, q_ in ((yw, sm, sq), (yr, smr, sqr)):                       # the epilogue statistics are those of the stored values
        assert float((s_.cpu() - y.double().sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(s_.abs().max()))
        assert float((q_.cpu() - (y.double() ** 2).sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(q_.abs().max()))
    # run-to-run bit-identity
    assert torch.equal(yr, ops.conv3d_gcr_split_persample(s0, prep))


@pytest.mark.parametrize("B,dims,C0,Cout,scattered,with_partial", [(2, (8, 16, 16), 32, 32, False, False), (2, (8, 16, 16), 32, 32, True, False),
                                                                   (1, (8, 8, 8), 128, 32, False, False), (2, (16, 8, 24), 64, 64, False, False),
                                                                   (1, (16, 16, 16), 128, 32, True, False), (3, (24, 16, 8), 32, 96, False, False), (1, (8, 8, 40), 16, 32, False, False),
                                                                   (2, (8, 16, 16), 32, 32, False, True), (1, (16, 16, 16), 64, 64, False, True)])
def test_conv3d_winograd32_against_fp64(B, dims, C0, Cout, scattered, with_partial):
    """Winograd F(2,3)-along-x form of the 32-wide column-block layers (csrc/unet_wino32.hip through gn_conv3d_gcr_split_wino / _wino_partial; layers:
    components/unet3d.py:127-144,291,330): both operand forms against torch in fp64, next to the direct x-strip kernel and the fp32-MFMA kernel, with and
    without the polyphase partial of a decoder's first convolution.  The bar is the f16x2 contract: error <= 2x the fp32-MFMA kernel's.  Volumes of one
    to three 8 x 8 x 8 tiles per axis: every voxel sits on a face somewhere; chains of tiles cross samples and column blocks."""
    g = torch.Generator().manual_seed(C0 + Cout + dims[2])
    D, H, W = dims
    x = torch.randn(B, C0, D, H, W, generator=g)
    if scattered:
        x = x * (torch.rand(B, 1, D, H, W, generator=g) < 0.05)
    w = torch.randn(Cout, C0, 3, 3, 3, generator=g) / (27 * C0) ** 0.5
    gamma, beta = torch.rand(C0, generator=g) + 0.5, torch.randn(C0, generator=g)
    pre = F.conv3d(F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1)
    part = None
    if with_partial:
        part = torch.randn(B, D // 2, H // 2, W // 2, 8 * Cout, generator=g)
        pre = pre + part.double().view(B, D // 2, H // 2, W // 2, 2, 2, 2, Cout).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, Cout, D, H, W)
        part = part.to(DEV)
    ref = F.relu(pre)
    s0 = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    st = ops.channel_stats(s0)
    a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV), with_act_scale=True)
    a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma.to(DEV), beta.to(DEV))
    cl = lambda t: t.permute(0, 4, 1, 2, 3).cpu().double()
    err = lambda t: float((cl(t) - ref).abs().max())
    # the yardstick: the fp32-MFMA kernel (it takes no partial: measured on the same layer without one)
    ref0 = F.relu(pre) if not with_partial else F.relu(F.conv3d(F.group_norm(x.double(), 8, gamma.double(), beta.double(), eps=1e-5), w.double(), None, padding=1))
    e32 = float((cl(ops.conv3d_gcr(s0, None, a0, d0, ops.pack_conv_weight(w).to(DEV), Cout)) - ref0).abs().max())
    e_dir = err(ops.conv3d_gcr_split(s0, None, a, d, ops.pack_conv_weight_split(w, ops.SPLIT_F16X2).to(DEV), Cout, act_inv=inv, partial=part))
    pkw = ops.pack_conv_weight_split_wino(w).to(DEV)
    yw, (sm, sq, V) = ops.conv3d_gcr_split_wino(s0, a, d, pkw, Cout, act_inv=inv, with_stats=True, partial=part)
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino32pc_kernel<true>"
    prep = ops.conv_affine_pack(w.to(DEV).contiguous(), a0, d0, st, wino=True)
    yr, (smr, sqr, _) = ops.conv3d_gcr_split_persample(s0, prep, with_stats=True, partial=part)
    e_w, e_r = err(yw), err(yr)
    print(f"B={B} {dims} {C0}->{Cout} {'scattered' if scattered else 'dense'}{' +partial' if with_partial else ''}: err vs fp64: fp32-MFMA {e32:.2e}, x-strip f16x2 {e_dir:.2e}, "
          f"Winograd-32 literal {e_w:.2e}, affine-in-weights {e_r:.2e}")
    bar = 2 * max(e32, 2e-6)
    assert e_w <= bar and e_r <= bar
    for y, s_, q_ in ((yw, sm, sq), (yr, smr, sqr)):                       # the epilogue statistics are those of the stored values
        assert float((s_.cpu() - y.double().sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(s_.abs().max()))
        assert float((q_.cpu() - (y.double() ** 2).sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(q_.abs().max()))
    # run-to-run bit-identity
    assert torch.equal(yw, ops.conv3d_gcr_split_wino(s0, a, d, pkw, Cout, act_inv=inv, partial=part))
    assert torch.equal(yr, ops.conv3d_gcr_split_persample(s0, prep, partial=part))


def test_conv3d_winograd32_wave_specialised_kernel_is_bit_identical_to_the_plain_one(monkeypatch):
    """the default launch of the 32-wide Winograd layers is the wave-specialised kernel (csrc/unet_wino32pc.hip: four waves multiply two z-slices each, four stage the
    halo and fetch the weights); GARMENTNETS_WINO32_PC=0 selects the kernel every wave of which does both (csrc/unet_wino32.hip).  Same tile, same products in the same
    order per output: bit-identical outputs -- literal pack, per-sample packs, with the polyphase partial, chains crossing samples and column blocks; statistics equal
    to fp64 rounding (merged by fp64 atomics in hardware order)"""
    g = torch.Generator().manual_seed(5)
    for (B, D, H, W, C, Cout, with_partial) in ((3, 16, 24, 32, 64, 64, True), (2, 8, 16, 16, 128, 32, False), (2, 24, 8, 8, 32, 96, False)):
        x = torch.randn(B, D, H, W, C, generator=g).to(DEV)
        w = torch.randn(Cout, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        st = ops.channel_stats(x)
        a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma, beta, with_act_scale=True)
        a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma, beta)
        pk = ops.pack_conv_weight_split_wino(w).to(DEV)
        prep = ops.conv_affine_pack(w.to(DEV).contiguous(), a0, d0, st, wino=True)
        part = torch.randn(B, D // 2, H // 2, W // 2, 8 * Cout, generator=g).to(DEV) if with_partial else None
        res = {}
        for pc, name in (("1", "conv3d_split_wino32pc_kernel<true>"), ("0", "conv3d_split_wino32_kernel<true>")):
            monkeypatch.setenv("GARMENTNETS_WINO32_PC", pc)
            yl, stl_ = ops.conv3d_gcr_split_wino(x, a, d, pk, Cout, act_inv=inv, with_stats=True, partial=part)
            assert ops._lib.load().gn_last_kernel().decode() == name
            yr, str_ = ops.conv3d_gcr_split_persample(x, prep, with_stats=True, partial=part)
            res[pc] = (yl, stl_, yr, str_)
        monkeypatch.delenv("GARMENTNETS_WINO32_PC")
        assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][2], res["0"][2])
        for i in (1, 3):
            for s1, s0 in zip(res["1"][i][:2], res["0"][i][:2]):
                assert float((s1 - s0).abs().max()) <= 1e-12 * max(1.0, float(s0.abs().max()))


def test_conv3d_winograd32_shape_contract_and_occupancy_aware_launch():
    """(1) shapes outside the 32-wide Winograd kernel's contract are refused with GN_EINVAL (ValueError), never run; (2) its occupancy-aware launch (active
    list at the kernel's 8 x 8 x 8 granularity + border-class constants from its own dense launch over the 8^3 at-rest volume) is bit-identical to its
    dense launch, through SingleConv.run behind a scattered volume (reach 1)"""
    from garmentnets_amd.components.unet3d import SingleConv
    w = torch.randn(32, 32, 3, 3, 3)
    pk = ops.pack_conv_weight_split_wino(w).to(DEV)
    ones = lambda B, C: (torch.ones(B, C, device=DEV), torch.zeros(B, C, device=DEV))
    for dims, cin in (((12, 8, 8), 32), ((8, 8, 12), 32)):                   # not whole 8 x 8 x 8 tiles
        with pytest.raises(ValueError):
            ops.conv3d_gcr_split_wino(torch.zeros(1, *dims, cin, device=DEV), *ones(1, cin), pk, 32)
    with pytest.raises(ValueError):                                           # a partial goes with the 32-wide kernel only
        ops.conv3d_gcr_split_wino(torch.zeros(1, 8, 8, 8, 32, device=DEV), *ones(1, 32), ops.pack_conv_weight_split_wino(torch.randn(128, 32, 3, 3, 3)).to(DEV), 128,
                                  partial=torch.zeros(1, 4, 4, 4, 8 * 128, device=DEV))
    g = torch.Generator().manual_seed(22)
    B, G, C = 3, 64, 32
    conv = SingleConv(C, 32).to(DEV)
    conv.load_state_dict({k: S.synthetic_tensor("w32." + k, tuple(v.shape), 4).to(DEV) for k, v in conv.state_dict().items()})
    x = torch.zeros(B, G, G, G, C)
    n = 60
    idx = torch.randint(0, G, (B - 1, n, 3), generator=g)
    for b in range(B - 1):
        x[b, idx[b, :, 0], idx[b, :, 1], idx[b, :, 2]] = torch.randn(n, C, generator=g).abs() * 2.0
    flat = torch.cat([((b * G + idx[b, :, 0]) * G + idx[b, :, 1]) * G + idx[b, :, 2] for b in range(B - 1)]).to(torch.int32).to(DEV)
    xg = x.to(DEV)
    ar = AR.DEFAULT.replace(conv_mode=AR.SPLIT_F16X2, affine_in_weights=True, winograd=True, winograd32=True)
    y_d, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=False))
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino32pc_kernel<true>"
    y_s, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=True))
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino32pc_kernel<true>"
    assert torch.equal(y_s, y_d) and bool(torch.isfinite(y_d).all()) and float(y_d.abs().max()) > 0
    y_strip, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=False, winograd32=False))
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_strip_kernel<true>"
    assert float((y_strip - y_d).abs().max()) <= 2e-5 * max(1.0, float(y_d.abs().max()))       # two fp32-class roundings of the same layer


def test_conv3d_winograd_occupancy_aware_launch_and_shape_contract():
    """(1) the occupancy-aware launch of the Winograd kernel (active-tile list + border-class constants from the DIRECT form's 5^3 launch: away from
    the cells the operand is exactly zero in either form) is bit-identical to its dense launch; (2) shapes outside the kernel's contract are
    refused with GN_EINVAL (ValueError), never run"""
    from garmentnets_amd.components.unet3d import SingleConv
    g = torch.Generator().manual_seed(21)
    B, G, C = 4, 32, 32
    conv = SingleConv(C, 128).to(DEV)
    conv.load_state_dict({k: S.synthetic_tensor("wn." + k, tuple(v.shape), 4).to(DEV) for k, v in conv.state_dict().items()})
    x = torch.zeros(B, G, G, G, C)
    n = 40
    idx = torch.randint(0, G, (B - 1, n, 3), generator=g)
    for b in range(B - 1):
        x[b, idx[b, :, 0], idx[b, :, 1], idx[b, :, 2]] = torch.randn(n, C, generator=g).abs() * 2.0
    flat = torch.cat([((b * G + idx[b, :, 0]) * G + idx[b, :, 1]) * G + idx[b, :, 2] for b in range(B - 1)]).to(torch.int32).to(DEV)
    xg = x.to(DEV)
    ar = AR.DEFAULT.replace(conv_mode=AR.SPLIT_F16X2, affine_in_weights=True, winograd=True)
    y_d, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=False))
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino_kernel<true>"
    y_s, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=True))
    assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino_kernel<true>"
    assert torch.equal(y_s, y_d) and bool(torch.isfinite(y_d).all()) and float(y_d.abs().max()) > 0
    y_direct, _ = conv.run(xg, None, sparse=dict(flat=flat, reach=1), arith=ar.replace(sparse_first_conv=False, winograd=False))
    assert float((y_direct - y_d).abs().max()) <= 2e-5 * max(1.0, float(y_d.abs().max()))
    # shape contract
    w = torch.randn(128, 32, 3, 3, 3, generator=g) * 0.05
    pk = ops.pack_conv_weight_split_wino(w).to(DEV)
    for dims in ((8, 8, 12), (6, 8, 8), (8, 12, 8)):
        s0 = torch.randn(1, *dims, 32, generator=g).to(DEV)
        with pytest.raises(ValueError):
            ops.conv3d_gcr_split_wino(s0, torch.ones(1, 32, device=DEV), torch.zeros(1, 32, device=DEV), pk, 128)
    # (Cout = 64: the 32-wide column-block kernel of round 6 -- whole 8 x 8 x 8 tiles, Cin <= 128)
    assert ops.wino_supported(32, 64, (8, 8, 8)) and not ops.wino_supported(32, 64, (4, 8, 8)) and not ops.wino_supported(144, 64, (8, 8, 8))
    assert not ops.wino_supported(272, 128, (8, 8, 8)) and ops.wino_supported(256, 256, (4, 8, 8)) and not ops.wino_supported(32, 48, (8, 8, 8))


def test_conv3d_winograd_chain_length_does_not_change_a_bit(monkeypatch):
    """the Winograd kernel's workgroups walk CHAINS of tiles (csrc/unet_wino.hip: a tile's last slice stages the next tile's first; the
    epilogue statistics leave once per chain): outputs are bit-identical for every chain length -- 1 (one tile per workgroup), 2 / 3 (chains
    that cross from one sample into the next: the drain-and-restart path; Cout = 256: two column blocks interleaved in the item order), 16 --
    for the literal form, the per-sample packs and the occupancy-aware (active-list) launch; the statistics agree to fp64 rounding (they are
    merged by fp64 atomics in whatever order the hardware schedules them, as in every launch of these kernels)."""
    from garmentnets_amd.components.unet3d import SingleConv
    g = torch.Generator().manual_seed(77)
    cases = []
    # (Cout = 32 / 64: the 32-wide column-block kernel of round 6, csrc/unet_wino32.hip -- the same chain machinery at its 8 x 8 x 8 tile granularity)
    for (B, D, H, W, C, Cout) in ((3, 16, 32, 32, 32, 128), (2, 8, 24, 40, 64, 256), (3, 16, 32, 32, 32, 32), (2, 8, 24, 40, 64, 64)):
        x = torch.randn(B, D, H, W, C, generator=g).to(DEV)
        w = torch.randn(Cout, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        st = ops.channel_stats(x)
        a, d, inv = ops.groupnorm_affine(st, None, 8, 1e-5, gamma, beta, with_act_scale=True)
        a0, d0 = ops.groupnorm_affine(st, None, 8, 1e-5, gamma, beta)
        cases.append((x, a, d, inv, ops.pack_conv_weight_split_wino(w).to(DEV), ops.conv_affine_pack(w.to(DEV).contiguous(), a0, d0, st, wino=True), Cout))
    # occupancy-aware: a scattered 32^3 volume through SingleConv (active-tile list + class constants)
    Bs, G, C = 4, 32, 32
    conv = SingleConv(C, 128).to(DEV)
    conv.load_state_dict({k: S.synthetic_tensor("wc." + k, tuple(v.shape), 4).to(DEV) for k, v in conv.state_dict().items()})
    xs = torch.zeros(Bs, G, G, G, C)
    idx = torch.randint(0, G, (Bs, 60, 3), generator=g)
    for b in range(Bs):
        xs[b, idx[b, :, 0], idx[b, :, 1], idx[b, :, 2]] = torch.randn(60, C, generator=g).abs() * 2.0
    flat = torch.cat([((b * G + idx[b, :, 0]) * G + idx[b, :, 1]) * G + idx[b, :, 2] for b in range(Bs)]).to(torch.int32).to(DEV)
    xs = xs.to(DEV)
    ar = AR.DEFAULT.replace(conv_mode=AR.SPLIT_F16X2, affine_in_weights=True, winograd=True, sparse_first_conv=True)

    def run_all():
        out = []
        for x, a, d, inv, pk, prep, Cout in cases:
            out.append(ops.conv3d_gcr_split_wino(x, a, d, pk, Cout, act_inv=inv, with_stats=True))
            out.append(ops.conv3d_gcr_split_persample(x, prep, with_stats=True))
        out.append(conv.run(xs, None, sparse=dict(flat=flat, reach=1), arith=ar, with_stats=True))
        assert ops._lib.load().gn_last_kernel().decode() == "conv3d_split_wino_kernel<true>"
        return out

    monkeypatch.setenv("GARMENTNETS_WINO_CHAIN", "1")
    ref = run_all()
    for chain in ("2", "3", "16"):
        monkeypatch.setenv("GARMENTNETS_WINO_CHAIN", chain)
        for (y0, st0), (y1, st1) in zip(ref, run_all()):
            assert torch.equal(y0, y1), f"chain {chain}"
            for s0_, s1_ in zip(st0[:2], st1[:2]):
                assert float((s0_ - s1_).abs().max()) <= 1e-12 * max(1.0, float(s0_.abs().max())), f"chain {chain}"
    monkeypatch.delenv("GARMENTNETS_WINO_CHAIN")
    for (y0, _), (y1, _) in zip(ref, run_all()):                      # the launcher's own choice
        assert torch.equal(y0, y1)


CONV_ORDERS = ("gcr", "cr", "crg", "cl", "ce", "bcr", "cbr", "cgr")


@pytest.mark.parametrize("order", CONV_ORDERS)
def test_conv_layer_orders_against_reference_golden(golden_dir, order):
    """every create_conv layer order (components/unet3d.py:19-91), not only the shipped 'gcr': the HIP SingleConv on the reference module's golden
    input / parameters / output (tests/golden/make_golden_ref.py conv_orders_case).  'gcr' takes the fused f16x2 path, the others the fp32-MFMA
    convolution + gn_affine_act"""
    from test_oracle_golden import conv_order_state
    from garmentnets_amd.components.unet3d import SingleConv
    z = np.load(os.path.join(golden_dir, "ref_conv_orders.npz"))
    m = SingleConv(16, 32, order=order, num_groups=4)
    sd = conv_order_state(z, order)
    if "b" in order:
        sd["batchnorm.num_batches_tracked"] = torch.tensor(0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = torch.from_numpy(z["x"]).permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    with torch.no_grad():
        y, st = m.run(x, with_stats=True)
    got = y.permute(0, 4, 1, 2, 3).cpu().numpy()
    ref = z["y_" + order]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(st[0].cpu().numpy(), got.astype(np.float64).sum(axis=(2, 3, 4)), rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("order", ["crg", "bcr", "cl"])
def test_unet_other_layer_orders_against_oracle(order):
    """a whole Abstract3DUNet (encoder, max-pool, two-source decoder convs, final conv) in a non-default layer order against the oracle's
    restatement, which tests/test_oracle_golden.py pins to the reference's modules for these orders"""
    from garmentnets_amd.components.unet3d import Abstract3DUNet
    torch.manual_seed(5)
    net = Abstract3DUNet(in_channels=32, out_channels=16, f_maps=32, layer_order=order, num_groups=8, num_levels=2)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for name, prm in net.named_parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.05 if prm.dim() > 1 else 0.3) + (1.0 if name.endswith("norm.weight") else 0.0))
        for name, buf in net.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.2)
    net = net.eval()
    sd = {"u." + k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 32, 8, 8, 8, generator=g)
    with torch.no_grad():
        ref = P.unet3d(sd, dict(f_maps=32, layer_order=order, num_groups=8, num_levels=2), x, prefix="u").numpy()
        y = net.to(DEV)(x.to(DEV)).cpu().numpy()
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))


def test_trilinear_against_grid_sample():
    g = torch.Generator().manual_seed(2)
    vol = torch.randn(1, 24, 5, 7, 9, generator=g)
    q = torch.rand(1, 4000, 3, generator=g) * 1.1 - 0.05      # includes out-of-range -> border clamp
    q[0, 0] = 0.0
    q[0, 1] = 1.0
    ref = F.grid_sample(vol, (2.0 * q - 1.0).view(1, -1, 1, 1, 3), mode="bilinear", padding_mode="border", align_corners=True)
    ref = ref.view(24, -1).t()
    out = ops.trilinear_sample(vol[0].permute(1, 2, 3, 0).contiguous().to(DEV), query=q[0].to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dims,Q,C", [((8, 8, 8), 8, 32), ((6, 5, 7), 9, 128), ((16, 16, 16), 21, 64), ((32, 32, 32), 32, 128), ((5, 9, 3), 12, 32)])
def test_lattice_brick_sampler_is_bit_identical(dims, Q, C):
    """gn_trilinear_sample on whole lattice slabs (brick kernel: voxel bounding box staged in LDS) == the per-query kernel,
    bit for bit (forced by a chunk that does not start on a slab), and == F.grid_sample within rounding."""
    D, H, W = dims
    vol = torch.randn(D, H, W, C, generator=torch.Generator().manual_seed(Q + C)).to(DEV)
    n = Q * Q * Q
    brick = ops.trilinear_sample(vol, Q=Q, m0=0, M=n)
    per_query = ops.trilinear_sample(vol, Q=Q, m0=1, M=n - 1)
    assert torch.equal(brick[1:], per_query)
    part = ops.trilinear_sample(vol, Q=Q, m0=2 * Q * Q, M=3 * Q * Q)          # a chunk of slabs 2..4
    assert torch.equal(part, brick[2 * Q * Q:5 * Q * Q])
    gp = P.grid_points(Q).reshape(1, -1, 1, 1, 3)
    ref = F.grid_sample(vol.cpu().permute(3, 0, 1, 2)[None], 2.0 * gp - 1.0, mode="bilinear", padding_mode="border", align_corners=True)
    np.testing.assert_allclose(brick.cpu().numpy(), ref.view(C, -1).t().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dims,Q", [((16, 16, 16), 16), ((8, 12, 10), 21), ((5, 5, 5), 32), ((32, 32, 32), 48)])
def test_lattice_sampler_fused_into_the_decoder_is_bit_identical(dims, Q):
    """SURVEY K14 (gn_implicit_decode_lattice_split: the lattice sampler inside the decoder MLP kernel, the next tile's corner gathers under
    the current tile's MFMAs, no sampled-row buffer) against gn_trilinear_sample + gn_implicit_decode_split on the same lattice rows: bit for
    bit -- whole lattices whose size is not a multiple of the 128-row tile, lattices finer and coarser than the volume, border corners
    (the 5^3 volume: most queries clamp), a row range in the middle of the lattice, and with a per-garment input scale"""
    D, H, W = dims
    g = torch.Generator().manual_seed(Q + D)
    raw = []
    for i, (n, k) in enumerate(((256, 32), (256, 256), (1, 256))):
        raw.append((torch.randn(n, k, generator=g) * (2.0 / k) ** 0.5, torch.randn(n, generator=g) * 0.1,
                    (torch.rand(n, generator=g) + 0.5) if i < 2 else None, torch.randn(n, generator=g) * 0.1 if i < 2 else None))
    pack = ops.pack_decode_split(raw).to(DEV)
    vol = (torch.randn(D, H, W, 32, generator=g) * 1.5).to(DEV)
    n = Q * Q * Q
    xs = ops.decoder_input_scale((vol.double() ** 2).sum(dim=(0, 1, 2)).view(1, 32), D * H * W, pack.smax)[0]
    for m0, M, scale in ((0, n, None), (0, n, xs), (Q * Q + 5, min(n - Q * Q - 5, 1000), None)):
        rows = ops.trilinear_sample(vol, Q=Q, m0=m0, M=M)
        want = ops.implicit_decode_split(rows, pack, xscale=scale)
        got = torch.full((M, 1), float("nan"), device=DEV)
        ops.implicit_decode_lattice_split(vol, Q, pack, got, xscale=scale, m0=m0, M=M)
        assert torch.equal(got, want), (dims, Q, m0, M)


def test_pipeline_with_the_fused_lattice_sampler_equals_the_default(golden_dir):
    """Arith(fused_lattice=True) through volume_lattice_forward of the whole pipeline == the default two-kernel path, bit for bit"""
    gd = np.load(os.path.join(golden_dir, "ref_dress_g32.npz"))
    B, n, G, Q, seed, stride = [int(v) for v in gd["meta"]]
    model = _model(S.default_hparams(grid=G, reduce_method=str(gd["reduce_method"])), seed)
    x, pos, batch = S.synthetic_cloud(B, n, seed)
    data = Batch(sizes=[n] * B, x=x, pos=pos, batch=batch).to(DEV)
    u3 = model.unet3d_forward(model.pointnet2_forward(data))
    a = model.volume_lattice_forward(u3, Q)["pred_volume"]
    b = model.volume_lattice_forward(u3, Q, arith=model.arith.replace(fused_lattice=True))["pred_volume"]
    assert torch.equal(a, b)
    np.testing.assert_allclose(b[0].cpu().numpy(), gd["wnf_volume"], rtol=0, atol=TOL)


@pytest.fixture(params=["f16x2", "fp32"])
def decode_mode(request):
    return request.param


@pytest.mark.parametrize("out_ch,M", [(1, 1000), (3, 37), (1, 32), (3, 1)])
def test_fused_decoder_against_torch_and_unfused(out_ch, M, decode_mode):
    """gn_implicit_decode_split / gn_implicit_decode (3-layer MLP in one kernel, either arithmetic) vs F.grid_sample + the oracle
    MLP, and vs the per-layer path."""
    from garmentnets_amd.networks.conv_implicit_wnf import ImplicitWNFDecoder
    g = torch.Generator().manual_seed(out_ch * 100 + M)
    dec = ImplicitWNFDecoder((128, 256, 256, out_ch), batch_norm=True)
    sd = {k: S.synthetic_tensor("volume_decoder." + k, tuple(v.shape), seed=3) for k, v in dec.state_dict().items()}
    dec.load_state_dict(sd)
    dec = dec.to(DEV).eval()
    dec.arith = AR.DEFAULT.replace(decode_mode=decode_mode)
    vol = torch.randn(2, 128, 6, 5, 7, generator=g)
    q = torch.rand(2, M, 3, generator=g)
    q[:, 0] = 1.0
    ref = P.implicit_decoder({"d." + k: v for k, v in sd.items()}, "d", vol, q)
    out = dec(vol.to(DEV), q.to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    dec.fused = False
    out2 = dec(vol.to(DEV), q.to(DEV))
    np.testing.assert_allclose(out2.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    dec.fused = True
    lat = dec.decode_lattice(vol.to(DEV), 9)
    gp = P.grid_points(9).reshape(1, -1, 3).repeat(2, 1, 1)
    ref_lat = P.implicit_decoder({"d." + k: v for k, v in sd.items()}, "d", vol, gp)
    np.testing.assert_allclose(lat.reshape(2, -1, out_ch).cpu().numpy(), ref_lat.numpy(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("k0,nh", [(128, 256), (32, 256), (32, 512)])
@pytest.mark.parametrize("out_ch,M,bn", [(1, 70001, True), (3, 4100, True), (2, 129, False), (4, 128, True)])
def test_decoder_split_against_fp64(out_ch, M, bn, k0, nh):
    """gn_implicit_decode_split (two fp16 planes per operand on the matrix cores, activations chained through registers, persistent
    workgroups) against an fp64 evaluation and against the fp32-MFMA kernel on the same rows: at least as accurate."""
    g = torch.Generator().manual_seed(out_ch * 7 + M + k0 + nh)
    # k0 = 32: the first layer with the UNet's final 1x1x1 conv folded in; nh = 512: the reference class's default hidden width
    # (networks/conv_implicit_wnf.py:122), implicit_decode_split512_kernel
    dims = [k0, nh, nh, out_ch]
    raw, ref = [], None
    x = torch.randn(M, k0, generator=g) * 2.0
    x[0] = 0.0
    h = x.double()
    for i in range(3):
        w = torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5 * (0.3 if i == 1 else 1.0)
        b = torch.randn(dims[i + 1], generator=g) * 0.1
        sc = (torch.rand(dims[i + 1], generator=g) + 0.5) if bn else None
        sh = torch.randn(dims[i + 1], generator=g) * 0.1 if bn else None
        raw.append((w, b, sc, sh))
        h = torch.relu(h @ w.double().t() + b.double())
        if bn:
            h = h * sc.double() + sh.double()
    xin = ops.new_rows(M, k0, DEV)
    xin.copy_(x.to(DEV))
    out = ops.implicit_decode_split(xin, ops.pack_decode_split(raw).to(DEV))
    dv = lambda t: None if t is None else t.to(DEV)
    layers = tuple((ops.pack_kpair(w).to(DEV) if i < 2 else w.contiguous().to(DEV), b.to(DEV), dv(sc), dv(sh), dims[i + 1]) for i, (w, b, sc, sh) in enumerate(raw))
    out32 = ops.implicit_decode(None, layers, M=M, xin=xin)
    e_split, e_f32 = (out.cpu().double() - h).abs().max().item(), (out32.cpu().double() - h).abs().max().item()
    print(f"decoder [{k0},{nh},{nh},{out_ch}] M={M}: err vs fp64 split {e_split:.2e}, fp32-MFMA {e_f32:.2e}, |y| max {h.abs().max():.2f}")
    assert e_split <= max(2 * e_f32, 2e-6) and e_split <= 2e-5


@pytest.mark.parametrize("k0", [32, 128])
@pytest.mark.parametrize("case", ["x=1e5", "x=1e3", "x=1e-3", "x=1e-6", "w1 row x1000", "w2 rows x1e-4", "w1 x1e-3 all"])
def test_decoder_split_range_contract(case, k0):
    """the decoder's input is the UN-normalised ReLU output of the last conv, its hidden layers are not normalised either: inputs from
    1e-6 to 1e5 and heavy-tailed / tiny weight rows must keep fp32-class accuracy (run-time input scale from the volume statistics +
    static per-unit scales, csrc/decode_split.hip).  Error relative to each output's own scale, against fp64, next to the fp32 kernel."""
    g = torch.Generator().manual_seed(k0 + len(case))
    M, dims = 5000, [k0, 256, 256, 3]
    mag = float(case[2:]) if case.startswith("x=") else 1.0
    x = torch.relu(torch.randn(M, k0, generator=g)) * mag                   # ReLU outputs, like the pre-final volume
    raw, h = [], x.double()
    for i in range(3):
        w = torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5
        if case == "w1 row x1000" and i == 0:
            w[5] *= 1000.0
        if case == "w2 rows x1e-4" and i == 1:
            w[::2] *= 1e-4
        if case == "w1 x1e-3 all" and i == 0:
            w *= 1e-3
        b = torch.randn(dims[i + 1], generator=g) * 0.1 * (mag if i == 0 else 1.0)
        sc, sh = torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1
        raw.append((w, b, sc, sh))
        h = torch.relu(h @ w.double().t() + b.double()) * sc.double() + sh.double()
    xin = ops.new_rows(M, k0, DEV)
    xin.copy_(x.to(DEV))
    pack = ops.pack_decode_split(raw).to(DEV)
    sumsq = (x.double() ** 2).sum(dim=0, keepdim=True).to(DEV)
    xs = ops.decoder_input_scale(sumsq, M, pack.smax)
    assert float(torch.log2(xs[0, 0])) == round(float(torch.log2(xs[0, 0]))) and float(xs[0, 0] * xs[0, 1]) == 1.0
    dv = lambda t: None if t is None else t.to(DEV)
    layers = tuple((ops.pack_kpair(w).to(DEV) if i < 2 else w.contiguous().to(DEV), b.to(DEV), dv(sc), dv(sh), dims[i + 1]) for i, (w, b, sc, sh) in enumerate(raw))
    out = ops.implicit_decode_split(xin, pack, xscale=xs[0])
    ops.implicit_decode(None, layers, M=M, xin=xin, out=out, run_if=xs[0, 2:3])      # the gated fp32 twin: runs only for `unsafe` garments
    out = out.cpu().double()
    unsafe = bool(xs[0, 2] != 0)
    out32 = ops.implicit_decode(None, layers, M=M, xin=xin).cpu().double()
    if unsafe:      # biases dwarf weights x activations somewhere (x = 1e-6 next to O(0.1) biases, tiny weight rows): the device sent
        assert torch.equal(out, out32)      # the garment to the fp32 kernel -- same bits as calling it directly
    scale = h.abs().amax(dim=0).clamp_min(1e-30)
    e16, e32 = ((out - h).abs().amax(dim=0) / scale).max().item(), ((out32 - h).abs().amax(dim=0) / scale).max().item()
    print(f"decoder [{k0},256,256,3] {case}: input scale 2^{float(torch.log2(xs[0, 0])):.0f}{' (unsafe -> fp32 kernel)' if unsafe else ''}, rel err split {e16:.2e}, fp32-MFMA {e32:.2e}")
    assert torch.isfinite(out).all() and e16 <= max(2 * e32, 3e-6)


@pytest.mark.parametrize("k0,out_ch,M", [(32, 3, 5000), (32, 1, 777), (128, 3, 1300)])
def test_decoder_batch_entries_are_the_single_calls_row_for_row(k0, out_ch, M):
    """gn_trilinear_sample_batch / gn_implicit_decode_split_batch / gn_implicit_decode_batch (round 6: the surface queries of a whole batch in three
    launches, one row set per blockIdx.y) against the per-garment calls, bit for bit -- four row sets with input magnitudes from 1e-6 to 1e3, so that every
    set has its own input scale and one of them is sent to the gated fp32 twin by the device; M not a multiple of the 128-query tile"""
    g = torch.Generator().manual_seed(k0 + out_ch + M)
    B, dims = 4, [k0, 256, 256, out_ch]
    mags = [1.0, 1e3, 1e-6, 0.03]
    vol = torch.stack([torch.relu(torch.randn(6, 7, 5, k0, generator=g)) * m for m in mags]).to(DEV)
    q = torch.rand(B, M, 3, generator=g).to(DEV)
    raw = []
    for i in range(3):
        w = torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5
        raw.append((w, torch.randn(dims[i + 1], generator=g) * 0.1, torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1))
    pack = ops.pack_decode_split(raw).to(DEV)
    st = ops.channel_stats(vol)
    xs = ops.decoder_input_scale(st[1], st[2], pack.smax)
    assert len({float(v) for v in xs[:, 0]}) >= 3 and float(xs[:, 2].sum()) >= 1          # different scales; at least one unsafe set
    layers = tuple((ops.pack_kpair(w).to(DEV) if i < 2 else w.contiguous().to(DEV), b.to(DEV), sc.to(DEV), sh.to(DEV), dims[i + 1]) for i, (w, b, sc, sh) in enumerate(raw))
    rows = ops.trilinear_sample_batch(vol, q)
    out = torch.full((B, M, out_ch), 7.0, dtype=torch.float32, device=DEV)
    ops.implicit_decode_split_batch(rows, pack, out, xscale=xs)
    ops.implicit_decode_batch(rows, layers[:3], out, run_if=xs[:, 2:], run_if_stride=xs.stride(0))
    for b in range(B):
        r1 = ops.trilinear_sample(vol[b], query=q[b])
        assert torch.equal(rows[b], r1), b
        o1 = ops.implicit_decode_split(r1, pack, xscale=xs[b])
        ops.implicit_decode(None, layers, M=M, xin=r1, out=o1, run_if=xs[b, 2:3])
        assert torch.equal(out[b], o1), b
    assert torch.isfinite(out).all()
    o2 = torch.empty_like(out)                              # without scale records: every set through the split kernel, unscaled
    ops.implicit_decode_split_batch(rows[:2], pack, o2[:2])
    assert torch.equal(o2[0], ops.implicit_decode_split(rows[0], pack)) and torch.equal(o2[1], ops.implicit_decode_split(rows[1], pack))


def test_predict_falls_back_to_fp32_on_nan(monkeypatch):
    """a NaN in the WNF volume under the split-operand arithmetic (the only way a range violation can surface) makes predict_batch re-run
    the batch with the fp32 kernels"""
    from garmentnets_amd import predict as PR
    hp = S.default_hparams(grid=16, reduce_method="max")
    model = _model(hp, 3)
    x, pos, batch = S.synthetic_cloud(2, 1500, seed=5)
    data = Batch(sizes=[1500, 1500], x=x, pos=pos, batch=batch).to(DEV)
    want = PR.predict_batch(model, data, volume_size=24, auto_level=True, arith=model.arith.strict_fp32())
    orig, calls = ops.implicit_decode_split, {"n": 0}

    def poisoned(xin, pack, out=None, xscale=None):
        r = orig(xin, pack, out=out, xscale=xscale)
        calls["n"] += 1
        if calls["n"] == 1:
            r[0] = float("nan")
        return r
    monkeypatch.setattr(ops, "implicit_decode_split", poisoned)
    before = PR._FALLBACKS["count"]
    with pytest.warns(UserWarning):
        PR._FALLBACKS["count"] = 0
        got = PR.predict_batch(model, data, volume_size=24, auto_level=True)
    assert PR._FALLBACKS["count"] == 1
    PR._FALLBACKS["count"] += before
    for a, b in zip(got, want):
        assert torch.equal(a["wnf_volume"], b["wnf_volume"]) and torch.equal(a["faces"], b["faces"])


# ------------------------------------------------------------------------------------------------ isosurface
def _gpu_mc_raw(vol, level):
    v = torch.from_numpy(np.ascontiguousarray(vol, np.float32)).to(DEV)
    verts, faces, normals, values, counts = ops.mc33(v, level, 6 * vol.size + 16, 12 * vol.size + 16)
    nv, nf = [int(c) for c in counts.cpu()]
    return verts[:nv].cpu().numpy(), faces[:nf].cpu().numpy(), normals[:nv].cpu().numpy(), values[:nv].cpu().numpy()


def test_mc33_every_sign_pattern(golden_dir):
    g = np.load(os.path.join(golden_dir, "mc_golden.npz"))
    fo = vo = 0
    for i, vol in enumerate(g["cell_vols"]):
        nf, nv = int(g["cell_nf"][i]), int(g["cell_nv"][i])
        if i % 8 < 3:       # 3 of the 8 magnitude seeds per pattern on the GPU (each launch is ~10 kernels)
            v, f, n, a = _gpu_mc_raw(vol, 0.0)
            assert len(f) == nf and len(v) == nv, i
            assert np.array_equal(f, g["cell_faces"][fo:fo + nf]), i
            assert np.array_equal(v, g["cell_verts"][vo:vo + nv]), i
            assert np.array_equal(a, g["cell_values"][vo:vo + nv]), i
            np.testing.assert_allclose(n, g["cell_normals"][vo:vo + nv], atol=1e-6)
        fo += nf
        vo += nv


def test_mc33_all_cells_in_one_volume(golden_dir):
    """All 2048 golden cells + 3000 exact-level cells packed side by side (separated by all-negative padding) and
    triangulated in ONE launch; compared with the oracle on the same packed volume (bit exact)."""
    g = np.load(os.path.join(golden_dir, "mc_golden.npz"))
    for vols, level in ((g["cell_vols"], 0.0), (g["exact_vols"], 0.5)):
        n = len(vols)
        side = int(np.ceil(np.sqrt(n)))
        big = np.full((2, 3 * side, 3 * side), level - 1.0, np.float32)
        for i, c in enumerate(vols):
            r, q = divmod(i, side)
            big[:, 3 * r:3 * r + 2, 3 * q:3 * q + 2] = c
        ref = O.marching_cubes_raw(big, level)
        got = _gpu_mc_raw(big, level)
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0]) and np.array_equal(got[3], ref[3])
        np.testing.assert_allclose(got[2], ref[2], atol=1e-6)


@pytest.mark.parametrize("name", ["noise14", "smooth24", "aniso", "exact12", "shell32"])
def test_isosurface_volumes(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "mc_golden.npz"))
    vol, level = g[name + "_vol"], float(g[name + "_level"])
    Q = vol.shape[-1]
    sp = 1 / (Q - 1)
    v, f, n, a, vvox = MCU.marching_cubes(torch.from_numpy(vol).to(DEV), level, (sp,) * 3, "ascent")
    assert np.array_equal(f.cpu().numpy(), g[name + "_faces"])            # bit-exact cube / face indices
    assert v.dtype == torch.float64 and np.array_equal(v.cpu().numpy(), g[name + "_verts"])
    assert np.array_equal(a.cpu().numpy(), g[name + "_values"])
    np.testing.assert_allclose(n.cpu().numpy(), g[name + "_normals"], atol=1e-6)
    ggm = ops.ggm3d(torch.from_numpy(vol).to(DEV), 0.5)
    assert np.array_equal(ggm.cpu().numpy(), g[name + "_ggm"])            # scipy-compatible accumulation order
    assert np.array_equal(ops.gather_nn(ggm, vvox, sp).cpu().numpy(), g[name + "_verts_ggm"])
    assert np.array_equal(ops.scale_verts(vvox, sp).cpu().numpy(), g[name + "_verts"].astype(np.float32))
    desc = MCU.marching_cubes(torch.from_numpy(vol).to(DEV), level, (sp,) * 3, "descent")[1]
    assert np.array_equal(desc.cpu().numpy(), np.fliplr(g[name + "_faces"]))


@pytest.mark.parametrize("Q", [128, 256])
def test_isosurface_full_size_shell(golden_dir, Q):
    """BASELINE sizes: 128^3 against scikit-image checksums, 128^3 / 256^3 against the oracle + mesh invariants."""
    g = np.load(os.path.join(golden_dir, "mc_golden.npz"))
    vol = S.shell_volume(Q)
    r = MCU.wnf_to_mesh_gpu(torch.from_numpy(vol).to(DEV), 0.5, 0.5)
    f = r["faces"].cpu().numpy()
    v32 = (r["verts"].cpu().numpy()).astype(np.float32)
    if Q == 128 and np.array_equal(_sha(vol), g["shell128_vol_sha"]):
        assert len(v32) == int(g["shell128_nv"]) and len(f) == int(g["shell128_nf"])
        assert np.array_equal(_sha(f), g["shell128_faces_sha"])
        assert np.array_equal(_sha(v32), g["shell128_verts_sha"])
        assert np.array_equal(_sha(r["volume_value"].cpu().numpy()), g["shell128_values_sha"])
        assert np.array_equal(_sha(r["ggm"].cpu().numpy()), g["shell128_ggm_sha"])
    ov, of, on, oa = O.marching_cubes(vol, 0.5, (1 / (Q - 1),) * 3)
    assert np.array_equal(f, of) and np.array_equal(r["verts"].cpu().numpy(), ov)
    np.testing.assert_allclose(r["normals"].cpu().numpy(), on, atol=1e-6)
    # size-independent properties: closed 2-manifold (every edge shared by exactly two faces), every vertex used,
    # Euler characteristic of a torus-free closed surface family: V - E + F even
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert np.all(counts == 2)
    assert len(np.unique(f)) == len(ov)
    assert (len(ov) - len(counts) + len(f)) % 2 == 0


def test_mc_error_contract():
    vol = torch.zeros(4, 4, 4, device=DEV)
    with pytest.raises(ValueError):
        MCU.marching_cubes(vol, 0.5)
    vol[1, 1, 1] = 1.0
    with pytest.raises(RuntimeError):
        MCU.marching_cubes(vol, 1.0)
    with pytest.raises(ValueError):
        MCU.marching_cubes(torch.zeros(1, 4, 4, device=DEV), 0.0)


# ------------------------------------------------------------------------------------------------ end to end
def test_predict_end_to_end_against_oracle():
    """predict.py:138-209 on the GPU vs the oracle on the same inputs (dress cloud, G=32/max, Q=32)."""
    from garmentnets_amd.predict import predict_batch
    hp = S.default_hparams(grid=32, reduce_method="max")
    sd = S.synthetic_state_dict(hp, 0)
    x, pos, batch = S.synthetic_cloud(1, 6000, seed=0)
    ref = P.predict(sd, hp, x, pos, batch, Q=32, level=0.5, sigma=0.5)["garments"][0]
    model = _model(hp, 0)
    out = predict_batch(model, Batch(sizes=[6000], x=x, pos=pos, batch=batch).to(DEV), volume_size=32, iso_surface_level=0.5,
                        gradient_sigma=0.5)[0]
    wnf = out["wnf_volume"].cpu().numpy()
    np.testing.assert_allclose(wnf, ref["wnf_volume"], rtol=0, atol=TOL)
    # grip-point post-processing (predict.py:254-274)
    assert np.array_equal(out["pred_global_nocs_grip_point"].cpu().numpy(), ref["pred_global_nocs_grip_point"])
    assert np.array_equal(out["pred_nocs_grip_point"].cpu().numpy(), ref["pred_nocs_grip_point"])
    np.testing.assert_allclose(out["pred_global_confidence"].cpu().numpy(), ref["pred_global_confidence"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["global_feature"].cpu().numpy(), ref["global_feature"], rtol=0, atol=TOL)
    # the isosurface of the GPU volume is bit-exact w.r.t. the oracle run on that same volume
    iso = P.isosurface(wnf, 0.5, 0.5)
    assert np.array_equal(out["faces"].cpu().numpy(), iso["faces"])
    assert np.array_equal(out["verts"].cpu().numpy(), iso["verts"])
    assert np.array_equal(out["volume_gradient_magnitude"].cpu().numpy(), iso["verts_ggm"])
    # and close to the oracle's own end-to-end mesh (same topology unless a WNF value sits within 1e-4 of the level)
    if np.array_equal(out["faces"].cpu().numpy(), ref["faces"]):
        np.testing.assert_allclose(out["verts"].cpu().numpy(), ref["verts"], atol=5e-3)
        np.testing.assert_allclose(out["warp_field"].cpu().numpy(), ref["warp_field"], atol=5e-3)
    sq = torch.from_numpy(iso["verts"].astype(np.float32)).view(1, -1, 3)
    vol_t = model.unet3d_forward(model.pointnet2_forward(Batch(sizes=[6000], x=x, pos=pos, batch=batch).to(DEV)))
    warp_ref = P.implicit_decoder(sd, "surface_decoder", vol_t["out_feature_volume"].cpu().contiguous(), sq).view(-1, 3)
    np.testing.assert_allclose(out["warp_field"].cpu().numpy(), warp_ref.numpy(), rtol=0, atol=TOL)


@pytest.mark.parametrize("reduce", ["max", "mean"])
def test_run_twice_is_bit_identical(reduce):
    """determinism: two runs of the whole dense path on the same input give the same bits.  max-scatter is order-independent by
    construction, mean-scatter accumulates exact fp64 partial sums (csrc/grid.hip), the GroupNorm statistics are fp64 sums of fp32
    partials (exact, hence order-independent, for any realistic exponent spread)."""
    hp = S.default_hparams(grid=32, reduce_method=reduce)
    model = _model(hp, 2)
    x, pos, batch = S.synthetic_cloud(3, 2000, seed=9)
    data = Batch(sizes=[2000] * 3, x=x, pos=pos, batch=batch).to(DEV)
    runs = []
    for _ in range(2):
        with torch.no_grad():
            p2 = model.pointnet2_forward(data)
            vin = model.volume_agg(p2["nocs_data"]).clone()
            u3 = model.unet3d_forward(p2)
            wnf = model.volume_lattice_forward(u3, 40)["pred_volume"].clone()
        runs.append((p2["per_point_logits"].clone(), p2["global_feature"].clone(), vin, u3.pre_final.clone(), wnf))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_mean_scatter_many_points_per_cell_is_exact():
    """1500 points into 3 cells (the collapsed clouds of the seeded synthetic weights): the mean equals the fp64 mean rounded through
    fp32(sum) / count, whatever the arrival order"""
    g = torch.Generator().manual_seed(0)
    N, C = 1500, 128
    src = (torch.randn(N, C, generator=g) * torch.logspace(-3, 3, C)).float()
    flat = torch.randint(0, 3, (N,), generator=g).to(torch.int32) * 1000 + 17
    vol = ops.grid_scatter(src.to(DEV), flat.to(DEV), 1, (16, 16, 16), "mean").reshape(-1, C).cpu()
    for cell in flat.unique().tolist():
        sel = flat == cell
        want = (src[sel].double().sum(dim=0).float() / float(sel.sum())).float()
        assert torch.equal(vol[cell], want)
    assert int((vol != 0).any(dim=1).sum()) == 3


# ------------------------------------------------------------------------------------------------ widening: metrics
def test_chamfer_against_ckdtree():
    """eval.py:259-271,381-402 with scipy's cKDTree (the reference's own dependency) as the oracle"""
    from scipy.spatial import cKDTree
    from garmentnets_amd.common import metrics
    rng = np.random.default_rng(5)
    pred = rng.random((3001, 3)).astype(np.float32)
    gt = (rng.random((2500, 3)) * 1.1).astype(np.float32)
    pred_sim, gt_sim = rng.normal(size=(3001, 3)).astype(np.float32), rng.normal(size=(2500, 3)).astype(np.float32)
    fd, fi = cKDTree(gt).query(pred, k=1)
    bd, bi = cKDTree(pred).query(gt, k=1)
    idx, d2 = ops.nearest_neighbor(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt).to(DEV))
    assert np.array_equal(idx.cpu().numpy(), fi)
    np.testing.assert_allclose(np.sqrt(d2.cpu().numpy().astype(np.float64)), fd, rtol=1e-5, atol=1e-7)
    t = lambda a: torch.from_numpy(a).to(DEV)
    c = metrics.chamfer(t(pred), t(gt))
    assert abs(float(c["chamfer_symmetrical"]) - np.mean([fd.mean(), bd.mean()])) < 1e-7
    h = metrics.hybrid_chamfer(t(pred), t(gt), t(pred_sim), t(gt_sim))
    ref_f = np.linalg.norm(pred_sim - gt_sim[fi], axis=1).mean()
    ref_b = np.linalg.norm(gt_sim - pred_sim[bi], axis=1).mean()
    assert abs(float(h["hybrid_chamfer_forward"]) - ref_f) < 1e-6 and abs(float(h["hybrid_chamfer_backward"]) - ref_b) < 1e-6


# ------------------------------------------------------------------------------------------------ fused set abstraction
@pytest.mark.parametrize("level", ["sa1", "sa2", "3-32-64-128", "3-64-128-128", "3-32-32-64", "3-64-64-64", "64-64-64-128", "64-64-128-256", "128-128-128-128",
                                   "128-128-256-256", "0-64-64-128"])
@pytest.mark.parametrize("self_loops", [True, False])
def test_sa_fused_against_unfused_chain(level, self_loops):
    """gn_sa_fused (gather -> edge MLP on the matrix cores -> BatchNorm -> max in one kernel) against the unfused chain
    gn_sa_gather -> gn_linear x3 -> gn_segment_max on the same fps / ball-query tables: same exact-fp32 products, another summation
    order -> 1e-5 relative; ragged batch, empty balls, balls beyond the 64-neighbour cap, centres past the last full group."""
    from garmentnets_amd.components import pointnet2 as PN
    from garmentnets_amd.components.mlp import MLP
    sizes = [2500, 37, 1300, 1]
    x0, pos, batch = _ragged_cloud(sizes, 77)
    g = torch.Generator().manual_seed(3)
    if level == "sa1":
        cin, dims, ratio, rad = 3, [6, 64, 64, 128], 0.5, 0.05
        x = x0
    elif level == "sa2":
        cin, dims, ratio, rad = 128, [131, 128, 128, 256], 0.25, 0.1
        x = torch.randn(pos.shape[0], 128, generator=g)
    else:        # round 5: the other instantiated edge MLPs (SA_SHAPES in csrc/sa_fused.hip), incl. a position-only one (no point features)
        c, n1, n2, n3 = [int(v) for v in level.split("-")]
        assert ops.sa_fused_supported(c, [n1, n2, n3])
        cin, dims, ratio, rad = c, [c + 3, n1, n2, n3], 0.5, 0.07
        x = x0 if c == 3 else (torch.randn(pos.shape[0], c, generator=g) if c else None)
    mod = PN.SAModule(ratio, rad, MLP(dims, batch_norm=True))
    sd = {k: S.synthetic_tensor("sa." + k, tuple(v.shape), 5) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    mod = mod.to(DEV).eval()
    mod.conv.add_self_loops = self_loops
    xin = None
    if x is not None:
        xin = ops.new_rows(x.shape[0], cin, DEV)
        xin.copy_(x.to(DEV))
    seg = Segments(sizes, DEV)
    try:
        saved, PN.FUSED_SA = PN.FUSED_SA, True
        out_f, cpos_f, cseg = mod(xin, pos.to(DEV), seg)
        PN.FUSED_SA = False
        out_u, cpos_u, _ = mod(xin, pos.to(DEV), seg)
    finally:
        PN.FUSED_SA = saved
    assert out_f.shape == out_u.shape == (cseg.total, dims[-1]) and torch.equal(cpos_f, cpos_u)
    err = float((out_f - out_u).abs().max())
    scale = float(out_u.abs().max())
    print(f"{level} self_loops={self_loops}: fused vs unfused max err {err:.2e} (max |y| {scale:.2f}, {cseg.total} centres)")
    assert err <= 1e-5 * max(1.0, scale)
    assert torch.equal(out_f == 0, out_u == 0) or err <= 1e-6


def test_sa_fused_runs_in_the_pipeline(golden_dir):
    """the fused kernel is what pointnet2_forward launches for the shipped hyper-parameters (both SA levels)"""
    hp = S.default_hparams(grid=16)
    model = _model(hp, 0)
    assert model.pointnet2_nocs.sa1_module._fused_pack() is not None and model.pointnet2_nocs.sa2_module._fused_pack() is not None
    assert ops.sa_fused_supported(3, [64, 64, 128]) and ops.sa_fused_supported(128, [128, 128, 256]) and not ops.sa_fused_supported(5, [64, 64, 128])


# ------------------------------------------------------------------------------------------------ occupancy-aware first convolution
@pytest.mark.parametrize("G,mode,aiw", [(32, 4, True), (36, 4, True), (32, 4, False), (36, 4, False), (20, 2, False)])
def test_sparse_first_conv_is_bit_identical_to_dense(G, mode, aiw):
    """the first TWO UNet convolutions behind a scattered volume (the encoder's first DoubleConv, 128 -> 128 -> 32): only the tiles that
    can see an occupied cell (within 1 voxel for the first layer, 2 for the second) go through the matrix cores, the rest are
    border-class constants (csrc/unet_split.hip tile_active / kconst / kreach).  Both outputs must equal the dense launches bit for
    bit (the epilogue statistics up to the order of their fp64 atomics): occupied cells in corners / on faces / in the interior, a
    garment without any point, a grid that is not a multiple of the tile."""
    from garmentnets_amd.components.unet3d import DoubleConv
    g = torch.Generator().manual_seed(G)
    B, C = 3, 128
    cells = [torch.tensor([[0, 0, 0], [G - 1, G - 1, G - 1], [0, G - 1, 5], [G // 2, G // 2, G // 2], [G // 2, G // 2, G // 2 + 1], [3, 8, 9], [4, 7, 8]]),
             torch.zeros((0, 3), dtype=torch.int64),
             torch.randint(0, G, (200, 3), generator=g)]
    flat = torch.cat([(((b * G + c[:, 0]) * G + c[:, 1]) * G + c[:, 2]) for b, c in enumerate(cells)]).to(torch.int32)
    feats = torch.randn(flat.numel(), C, generator=g)
    vol, stats = ops.grid_scatter(feats.to(DEV), flat.to(DEV), B, (G, G, G), "max", with_stats=True)
    dc = DoubleConv(C, 32, encoder=True)                                   # 128 -> 128 -> 32, the shipped encoders.0
    dc.load_state_dict({k: S.synthetic_tensor("c." + k, tuple(v.shape), 1) for k, v in dc.state_dict().items()})
    dc = dc.to(DEV)
    # aiw: both layers in the affine-in-weights form (arith.affine_in_weights: per-sample weight packs + bias table, csrc/conv_prep.hip) --
    # the occupancy-aware launch must equal ITS dense launch bit for bit just the same
    a_sp = AR.DEFAULT.replace(conv_mode=mode, sparse_first_conv=True, affine_in_weights=aiw)
    a_dn = a_sp.replace(sparse_first_conv=False)
    sp1 = dict(flat=flat.to(DEV), reach=1)
    y1_s, st1_s = dc.SingleConv1.run(vol, None, stats, None, sparse=sp1, arith=a_sp)
    assert ("rest_out" in sp1) == aiw
    y2_s, st2_s = dc.SingleConv2.run(y1_s, None, st1_s, sparse=dict(flat=flat.to(DEV), reach=2, small_in=sp1["small_out"], rest_in=sp1.get("rest_out")),
                                     arith=a_sp)
    both_s, _ = dc.run(vol, None, stats, None, sparse_flat=flat.to(DEV), arith=a_sp)
    d1 = dict(flat=flat.to(DEV), reach=1)
    y1_d, st1_d = dc.SingleConv1.run(vol, None, stats, None, sparse=d1, arith=a_dn)
    assert "small_out" not in d1
    # (layer 2's GroupNorm affine from the SAME statistics in both forms: the two launches' fp64 atomic sums agree to 1 ulp only, which
    #  once in a while lands on the other side of an fp32 rounding boundary of the affine -- that is the atomics' order, not the kernels)
    y2_d, st2_d = dc.SingleConv2.run(y1_d, None, st1_s, sparse=dict(flat=flat.to(DEV), reach=2, rest_in=d1.get("rest_out")), arith=a_dn)
    if aiw:            # the garment without a point is at rest everywhere: every interior voxel of layer 1's output IS the rest value
        assert torch.equal(y1_d[1, 3, 4, 5], d1["rest_out"][1]) and torch.equal(y1_d[1, G // 2, G // 2, G // 2], d1["rest_out"][1])
    f1, f2 = ops.grid_tile_flags(flat.to(DEV), B, (G, G, G), 1), ops.grid_tile_flags(flat.to(DEV), B, (G, G, G), 2)
    a1, a2 = f1.sum(dim=1).tolist(), f2.sum(dim=1).tolist()
    print(f"G={G} mode={mode}: active tiles per garment, layer 1 {a1} / layer 2 {a2} of {f1.shape[1]}")
    assert a1[1] == 0 and a2[1] == 0 and 0 < a1[0] <= a2[0] < f1.shape[1] and bool((f2 >= f1).all())
    assert torch.equal(y1_s, y1_d) and torch.equal(y2_s, y2_d) and torch.equal(both_s, y2_d)
    # the statistics are fp64 atomic sums of identical per-tile fp32 partials: equal up to the order of the fp64 additions (1 ulp)
    for (s_s, q_s, _), (s_d, q_d, _) in ((st1_s, st1_d), (st2_s, st2_d)):
        assert float(((s_s - s_d).abs() / s_d.abs().clamp_min(1e-30)).max()) <= 1e-13 and float(((q_s - q_d).abs() / q_d.abs().clamp_min(1e-30)).max()) <= 1e-13
    assert bool(torch.isfinite(y2_s).all()) and float(y2_s.abs().max()) > 0


# ------------------------------------------------------------------------------------------------ GroupNorm affine in per-sample weights
@pytest.mark.parametrize("Cin,Cout,dims,B,kind", [(32, 128, (32, 32, 32), 4, "scattered"), (32, 64, (64, 64, 32), 2, "scattered"),
                                                  (32, 32, (64, 64, 64), 1, "scattered"), (16, 32, (5, 7, 9), 2, "dense"),
                                                  (128, 32, (16, 16, 16), 2, "offset"), (64, 96, (12, 20, 9), 3, "dense"),
                                                  (32, 32, (64, 64, 64), 1, "offset")])
def test_affine_in_weights_conv_against_fp64(Cin, Cout, dims, B, kind):
    """GroupNorm -> Conv3d -> ReLU (components/unet3d.py:66-76) with the affine folded into per-sample weights and a border-class bias table
    (gn_conv_affine_pack + gn_conv3d_gcr_split_persample): exact algebra for ANY rest value c, so it is checked on a scattered volume
    (c = 0), on a volume at a per-channel offset with a few cells disturbed (c = that offset) and on dense noise (c = 0, nothing at
    rest), against torch in fp64 and against the standard f16x2 form.  Shapes chosen to reach every kernel variant (128-wide, 64-wide,
    x-strip, 32-wide with ragged tiles)."""
    from garmentnets_amd.components.unet3d import SingleConv
    g = torch.Generator().manual_seed(Cin * 7 + Cout + dims[0])
    D, H, W = dims
    rest = None
    if kind == "dense":
        x = torch.randn(B, D, H, W, Cin, generator=g) * 1.7 + 0.3
    else:
        x = torch.zeros(B, D, H, W, Cin)
        if kind == "offset":
            rest = torch.rand(B, Cin, generator=g) * 2.0
            rest[:, ::5] = 0.0                                    # ReLU outputs: some channels rest at zero
            x = x + rest[:, None, None, None, :]
        n = max(8, D * H * W // 400)
        for b in range(B - 1):                                    # the last garment has no point at all
            idx = torch.stack([torch.randint(0, D, (n,), generator=g), torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g)], 1)
            idx[0] = torch.tensor([0, 0, 0]); idx[1] = torch.tensor([D - 1, H - 1, W - 1])
            x[b, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.randn(n, Cin, generator=g).abs() * 3.0
    conv = SingleConv(Cin, Cout)
    conv.load_state_dict({k: S.synthetic_tensor("aw." + k, tuple(v.shape), 3) for k, v in conv.state_dict().items()})
    xd = x.permute(0, 4, 1, 2, 3).double()
    ref = F.relu(F.conv3d(F.group_norm(xd, conv.groupnorm.num_groups, conv.groupnorm.weight.double(), conv.groupnorm.bias.double(), eps=1e-5),
                          conv.conv.weight.double(), None, padding=1)).permute(0, 2, 3, 4, 1)
    conv = conv.to(DEV)
    xg = x.to(DEV)
    fake_flat = torch.zeros((1,), dtype=torch.int32, device=DEV)
    a_new = AR.DEFAULT.replace(conv_mode=4, sparse_first_conv=False, affine_in_weights=True)
    sp = dict(flat=fake_flat, reach=1) if rest is None else dict(flat=fake_flat, reach=2, rest_in=rest.to(DEV))
    y_new, (s_new, q_new, V) = conv.run(xg, None, sparse=sp, arith=a_new)
    kern = ops._lib.load().gn_last_kernel().decode()
    assert "rest_out" in sp
    y_old, (s_old, q_old, _) = conv.run(xg, None, arith=a_new.replace(affine_in_weights=False))
    scale = float(ref.abs().max())
    e_new, e_old = float((y_new.cpu().double() - ref).abs().max()), float((y_old.cpu().double() - ref).abs().max())
    print(f"{Cin}->{Cout} {dims} B={B} {kind}: err vs fp64 affine-in-weights {e_new:.2e} / standard {e_old:.2e} (max |y| {scale:.2f}); kernel {kern}")
    assert e_new <= 2e-5 * max(1.0, scale) and e_new <= 2 * max(e_old, 2e-6 * max(1.0, scale))
    # the epilogue statistics are those of the stored values
    assert float((s_new.cpu() - y_new.double().sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(s_new.abs().max()))
    assert float((q_new.cpu() - (y_new.double() ** 2).sum(dim=(1, 2, 3)).cpu()).abs().max()) <= 1e-9 * max(1.0, float(q_new.abs().max()))
    if kind != "dense" and min(dims) >= 12:
        # a garment at rest: interior voxels hold exactly the rest value this layer reports for the next one
        assert torch.equal(y_new[B - 1, D // 2, H // 2, W // 2], sp["rest_out"][B - 1])
    assert bool(torch.isfinite(y_new).all())


def test_affine_in_weights_pipeline_equals_standard_form_to_fp32_class(golden_dir):
    """the whole dense phase with and without arith.affine_in_weights: the feature volume and the WNF agree to fp32-class error (the two
    forms round differently, neither is the reference), occupancy-aware or not"""
    hp = S.default_hparams(grid=32)
    model = _model(hp, 3)
    x, pos, batch = S.synthetic_cloud(2, 3000, 5)
    p2 = model.pointnet2_forward(Batch(sizes=[3000] * 2, x=x, pos=pos, batch=batch).to(DEV))
    outs = {}
    for aiw in (True, False):
        for sparse in (True, False):
            ar = AR.DEFAULT.replace(affine_in_weights=aiw, sparse_first_conv=sparse)
            outs[(aiw, sparse)] = model.unet3d_forward(p2, arith=ar)["out_feature_volume"].float().clone()
    assert torch.equal(outs[(True, True)], outs[(True, False)]) and torch.equal(outs[(False, True)], outs[(False, False)])
    scale = float(outs[(False, False)].abs().max())
    err = float((outs[(True, False)] - outs[(False, False)]).abs().max())
    print(f"feature volume: affine-in-weights vs standard {err:.2e} (max |v| {scale:.2f})")
    assert err <= 2e-5 * max(1.0, scale)


def test_polyphase_decoder_conv_with_the_skip_connection_at_rest():
    """the last decoder's first convolution, cat((skip, upsample(x))) -> gcr: the skip connection is encoder 0's output, at rest away from the
    cells -- its full-resolution launch takes the affine-in-weights form (SingleConv.run rest0=, the polyphase partial added in the same
    epilogue) and must agree with the literal polyphase form and with torch in fp64"""
    from garmentnets_amd.components.unet3d import SingleConv
    g = torch.Generator().manual_seed(77)
    B, C0, C1, Cout, D, H, W = 2, 32, 64, 32, 64, 64, 64
    rest = torch.rand(B, C0, generator=g) * 1.5
    rest[:, ::3] = 0.0
    x0 = torch.zeros(B, D, H, W, C0) + rest[:, None, None, None, :]
    n = 700
    idx = torch.stack([torch.randint(0, D, (n,), generator=g), torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g)], 1)
    idx[0] = torch.tensor([0, 0, 0]); idx[1] = torch.tensor([D - 1, H - 1, W - 1])
    x0[0, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.randn(n, C0, generator=g).abs() * 2.0          # garment 1 stays at rest everywhere
    x1 = torch.randn(B, D // 2, H // 2, W // 2, C1, generator=g) * 1.5 + 0.2
    conv = SingleConv(C0 + C1, Cout)
    conv.load_state_dict({k: S.synthetic_tensor("pr." + k, tuple(v.shape), 4) for k, v in conv.state_dict().items()})
    up = F.interpolate(x1.permute(0, 4, 1, 2, 3).double(), size=(D, H, W), mode="nearest")
    cat = torch.cat((x0.permute(0, 4, 1, 2, 3).double(), up), dim=1)
    ref = F.relu(F.conv3d(F.group_norm(cat, conv.groupnorm.num_groups, conv.groupnorm.weight.double(), conv.groupnorm.bias.double(), eps=1e-5),
                          conv.conv.weight.double(), None, padding=1)).permute(0, 2, 3, 4, 1)
    conv = conv.to(DEV)
    s0, s1 = x0.to(DEV), x1.to(DEV)
    y_rest, (sr, qr, _) = conv.run(s0, s1, rest0=rest.to(DEV))
    kern = ops._lib.load().gn_last_kernel().decode()
    y_lit, (sl, ql, _) = conv.run(s0, s1)
    scale = float(ref.abs().max())
    e_rest, e_lit = float((y_rest.cpu().double() - ref).abs().max()), float((y_lit.cpu().double() - ref).abs().max())
    print(f"polyphase {C0}+{C1}->{Cout} at {D}^3: err vs fp64 with the skip at rest {e_rest:.2e} / literal {e_lit:.2e} (max |y| {scale:.2f}); kernel {kern}")
    assert not torch.equal(y_rest, y_lit)                                  # the other form really ran
    assert e_rest <= 2e-5 * max(1.0, scale) and e_rest <= 2 * max(e_lit, 2e-6 * max(1.0, scale))
    assert float((sr - sl).abs().max()) <= 1e-5 * max(1.0, float(sl.abs().max())) and float((qr - ql).abs().max()) <= 1e-5 * max(1.0, float(ql.abs().max()))
    # off: the flag routes back to the literal form bit for bit
    y_off, _ = conv.run(s0, s1, rest0=rest.to(DEV), arith=AR.DEFAULT.replace(affine_in_weights=False))
    assert torch.equal(y_off, y_lit)


# ------------------------------------------------------------------------------------------------ polyphase decoder convolutions
@pytest.mark.parametrize("C0,C1,Cout,dims,B", [(32, 64, 32, (16, 16, 16), 1), (64, 128, 64, (8, 16, 8), 2), (128, 256, 128, (8, 8, 8), 1),
                                               (32, 64, 32, (64, 64, 64), 2), (16, 32, 64, (4, 6, 10), 1)])
@pytest.mark.parametrize("mode", [4, 2])
def test_polyphase_upsampled_conv_equals_literal_form(C0, C1, Cout, dims, B, mode):
    """SingleConv on cat((skip, upsample_nearest(x))) (components/unet3d.py:291,330): the polyphase form (upsampled channels as a
    2x2x2-tap convolution per output parity class on the coarse volume, merged weights, added in the fine launch's epilogue) against the
    literal form (src1 read at half resolution in the halo stage) and against torch in fp64.  Exact algebra, another rounding order:
    both must be fp32-class (csrc/upconv.hip: one wave per parity class, the coarse halo staged once for all eight)."""
    from garmentnets_amd.components.unet3d import SingleConv
    g = torch.Generator().manual_seed(C0 + C1 + dims[0])
    D, H, W = dims
    x0 = torch.randn(B, C0, D, H, W, generator=g)
    x1 = torch.randn(B, C1, D // 2, H // 2, W // 2, generator=g) * 1.5 + 0.2
    conv = SingleConv(C0 + C1, Cout)
    conv.load_state_dict({k: S.synthetic_tensor("p." + k, tuple(v.shape), 2) for k, v in conv.state_dict().items()})
    up = F.interpolate(x1.double(), size=(D, H, W), mode="nearest")
    cat = torch.cat((x0.double(), up), dim=1)
    ref = F.relu(F.conv3d(F.group_norm(cat, conv.groupnorm.num_groups, conv.groupnorm.weight.double(), conv.groupnorm.bias.double(), eps=1e-5),
                          conv.conv.weight.double(), None, padding=1))
    conv = conv.to(DEV)
    s0 = x0.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    s1 = x1.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    outs = {}
    for poly in (True, False):
        y, (s_, q_, V) = conv.run(s0, s1, arith=AR.DEFAULT.replace(conv_mode=mode, polyphase_upconv=poly))
        outs[poly] = (y.permute(0, 4, 1, 2, 3).cpu().double(), s_.cpu(), q_.cpu(), ops._lib.load().gn_last_kernel().decode())
    scale = float(ref.abs().max())
    e_poly, e_lit = float((outs[True][0] - ref).abs().max()), float((outs[False][0] - ref).abs().max())
    print(f"{C0}+{C1}->{Cout} {dims} mode {mode}: err vs fp64 polyphase {e_poly:.2e} / literal {e_lit:.2e} (max |y| {scale:.2f}); main kernel {outs[True][3]}")
    tol = 2e-5 if mode == 4 else 3e-4
    assert e_poly <= tol * max(1.0, scale) and e_poly <= 2 * max(e_lit, 2e-6)
    # the epilogue statistics are those of the final values in both forms
    for i in (1, 2):
        assert float((outs[True][i] - outs[False][i]).abs().max()) <= 1e-3 * max(1.0, float(outs[False][i].abs().max())) * (1e-2 if mode == 4 else 1.0)
