#!/usr/bin/env python3
"""bench.py -- GarmentNets inference throughput on MI355X (BASELINE.json metric: garments/s end to end).

    python bench.py --gpus N --steps K --warmup W                       # default = BASELINE config[2]: full pipeline, B=16/GPU, G=128, Q=128
    python bench.py --workload full --volume-size 256 --batch 8         # config[4]: 256^3 WNF + marching cubes
    python bench.py --workload pointnet2 --batch 32                     # config[1]: PointNet++ NOCS forward only
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path (predict.py:138-209: PointNet++ -> gridding -> 3-D UNet -> (Q,Q,Q) WNF decode -> Gaussian
gradient magnitude -> Lewiner marching cubes -> surface decode) over one batch of synthetic garments.  The GLOBAL batch
(batch x world garments, one seed) is sharded contiguously over the ranks with parallel.shard_range: config[3] (128 garments over 8
GPUs) is literally what runs under --gpus 8.  Garments never move between GPUs; the only collective is an all-gather of per-rank
timings (RCCL).  Rank 0 prints ONE JSON line:

  value / ms_per_step   K timed steps, inputs resident in HBM, results left on the device (barrier + synchronize on both sides, MAX
                        over ranks) -- the default arithmetic (f16x2 operand split for the 3x3x3 convs and the decoder MLPs)
  strict_fp32           the same K steps with --conv-mode fp32 --decode-mode fp32 (v_mfma_f32_32x32x2_f32 everywhere), own roofline
  with_host_io          the same K steps including the H2D copy of the clouds and the D2H copy of every mesh (predict.to_host): the
                        metric as SURVEY.md 8d words it, one batch at a time
  two_in_flight         the same K batches through predict.PredictJob (batch k+1's dense path queued before batch k's tail is finished,
                        its PointNet++ on a front stream beside batch k's UNet); two_in_flight_with_host_io: the same with the host
                        copies inside the timed region (the host batch copied on the front stream, the meshes on the tail stream) --
                        SURVEY.md 8d's metric as the library would run it; `fraction_of_headline` says what the copies cost
  hbm_members.roofs     fp64 FMA rate and LDS read rate measured on this box (tools/dev/roof_burn): the ceilings the fp64 members of the tail
                        (GGM, marching cubes) are quoted against next to the HBM fraction
  roofline              dominant kernel by time: launches bracketed with HIP events on the launch stream during the timed steps,
                        labelled with the kernel variant the C ABI reports having launched (gn_last_kernel), achieved = algorithmic
                        FLOPs (54*Cin*Cout per voxel) / time.  The kernel runs at the socket's power cap, so the line carries the
                        evidence: `sclk_mhz` / `socket_power_w` sampled from the GPU's hwmon nodes DURING the headline pass and
                        `frac_at_2400mhz` = frac x 2400 / sclk (what the same issue rate would be at the guide's peak clock).
                        `traffic` = HBM bytes per launch measured IN THIS RUN: two short rocprofv3 child passes of the same step on
                        this box (--pmc FETCH_SIZE, --pmc WRITE_SIZE, gfx950-corrected as tools/pmc_summary.py documents), or null
                        when rocprofv3 is not usable -- never a number from another box's committed profile
  occupancy_aware       the same K steps with the library's default occupancy-aware first two UNet convolutions (exact: bit-identical outputs,
                        tests/test_gpu_parity.py::test_sparse_first_conv_is_bit_identical_to_dense).  The HEADLINE value is measured with
                        that path switched OFF: every tile goes through the matrix cores.  The MAC count then does not depend on where
                        the points fall -- the SPEED still does: with Arith.affine_in_weights (on for the headline) the MFMA operand of
                        the first two encoder convolutions (73 % of the dominant kernel's FLOPs) is exactly zero outside the occupied
                        cells' neighbourhood (>= 99 % of the voxels of this input); zero operands draw less power, the clock rises
                        under the cap and `roofline.frac` counts those MACs at full value.  Hence:
  literal_affine        the same K steps with Arith.affine_in_weights off (the GroupNorm shift inside the MFMA operand: every voxel
                        non-zero) -- the occupancy-INDEPENDENT figure, with its own roofline (`literal_affine.roofline`); quote it
                        next to the headline
  hbm_members           the HBM-bound members of the path (zero-fill, scatter, max-pool, lattice sampler, GGM, MC33 stages): HIP-event time,
                        algorithmic bytes, fraction of 8 TB/s
  validation            untimed: a batch of IDENTICAL garments (PointConv self-loop quirk off) must give the same WNF and mesh in the
                        first and the last slot -- garbage in the upper slots of the benchmark batch cannot go unnoticed
  cpu_baseline          the CPU oracle (torch-CPU port of the reference path) timed on this host on garment 0 OF THE BENCHMARK BATCH; its WNF
                        volume, NOCS bins and occupied cells are compared with slot 0 of the timed HIP result (oracle_check)

INPUT (--input planted, the default): the seeded global batch of synthetic.synthetic_cloud(colour="position") clouds and the seeded
synthetic checkpoint with synthetic.plant_nocs_path -- the NOCS head decodes the clouds' colour channels (:= the garment's normalised,
64-bin quantised positions), so the predicted NOCS coordinates spread over the garment's own shape: ~4900 occupied cells of the 128^3
grid per garment, the occupancy a trained PointNet++ produces, a well-conditioned input on which the north-star tolerance (1e-4 on the
WNF against the oracle) is enforced by tests/test_gpu_fullsize.py::test_bench_batch_against_oracle on THIS batch.  The checkpoint also
carries synthetic.plant_wnf_path: a carrier path through the UNet and the WNF decoder (multi-scale blur of the occupancy) that makes the
0.5 level set a thin shell around the garment -- a real garment's mesh size (tens of thousands of vertices) instead of the ~500 k-vertex
sponge random decoder weights give, so the iso-surface tail, the surface decoder and the host copies are timed on a garment-like amount of
work.  --input noisy_wnf: round 3's input (planted NOCS path only; the random-weight WNF as the tail's stress case).  --input collapsed:
rounds 1-2's input (uniform colours, un-planted random weights: every cloud collapses into ~5 cells; GroupNorm over a >99.99 % empty
volume amplifies rounding ~100x -- a conditioning study, not a headline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 (v_mfma_f32_32x32x16_*)
SPLIT_PRODUCTS = {"f16x2": 3, "bf16x3": 6, "bf16x2": 3}     # matrix-core products per fp32 product (csrc/unet_split.hip)
PEAK_HBM_GBS = 8000.0
CLOUD_SEED = 20260928


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="full", choices=["full", "pointnet2"],
                    help="full = BASELINE config[2]/[4] (whole predict path); pointnet2 = config[1] (PointNet2NOCS + NOCS post-processing only)")
    ap.add_argument("--batch", type=int, default=None, help="garments per GPU per step (default 16; 32 for --workload pointnet2)")
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--grid", type=int, default=128, help="feature-volume edge G (north_star: 128; reference ckpt default: 32)")
    ap.add_argument("--reduce", default="mean", choices=["mean", "max"])
    ap.add_argument("--volume-size", type=int, default=128, help="WNF query volume edge Q")
    ap.add_argument("--input", default="planted", choices=["planted", "noisy_wnf", "collapsed"],
                    help="planted (default): position-coloured clouds + the planted NOCS path (realistic occupancy, well conditioned) + the planted WNF "
                         "carrier (garment-like shell, a real garment's mesh size); noisy_wnf: round 3's input -- the planted NOCS path only, the WNF is "
                         "random-weight noise with ~10x a real garment's vertices (the stress case for the iso-surface tail); collapsed: "
                         "rounds 1-2's degenerate input (conditioning study)")
    ap.add_argument("--conv-mode", default="f16x2", choices=["f16x2", "fp32", "bf16x3", "bf16x2"],
                    help="arithmetic of the 3x3x3 convs of the HEADLINE pass: f16x2 (default; fp32 operands split into two fp16 planes, fp32 "
                         "accumulation), fp32 (v_mfma_f32_32x32x2_f32), bf16x3, bf16x2 (PREVIEW: does not guarantee the 1e-4 WNF tolerance -- "
                         "0.9-1.2e-4 observed on the G=32 goldens)")
    ap.add_argument("--winograd", default="on", choices=["on", "off"],
                    help="on (default): the 128-wide f16x2 convolutions over one full-resolution source -- above all the first encoder convolution -- in Winograd "
                         "F(2,3) form along x (36 instead of 54 matrix-core tap products per output pair; csrc/unet_wino.hip); off: the direct form everywhere")
    ap.add_argument("--winograd32", default="on", choices=["on", "off"],
                    help="on (default): the 32- / 64-wide layers of the two finest levels in the same Winograd form (csrc/unet_wino32.hip, round 6); off: the direct "
                         "x-strip kernel of rounds 2 - 5")
    ap.add_argument("--decode-mode", default="f16x2", choices=["f16x2", "fp32"], help="arithmetic of the decoder MLPs of the headline pass")
    ap.add_argument("--pipeline-depth", type=int, default=1, choices=[1, 2],
                    help="1 (default): one batch at a time (predict.predict_batch, the reference's loop); 2: every timed pass keeps two batches in "
                         "flight -- batch k+1's dense path is queued before batch k's host-synchronising tail is finished (predict.PredictJob).  "
                         "Either way a timed pass begins and finishes all its K batches")
    ap.add_argument("--no-in-flight-pass", action="store_true", help="skip the extra timed pass with two batches in flight (reported as two_in_flight)")
    ap.add_argument("--no-strict-pass", action="store_true", help="skip the second timed pass in strict fp32 arithmetic")
    ap.add_argument("--no-host-io-pass", action="store_true", help="skip the timed pass that includes H2D of the clouds / D2H of the meshes")
    ap.add_argument("--no-occupancy-pass", action="store_true", help="skip the timed passes with the occupancy-aware first convolution")
    ap.add_argument("--no-validate", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic on this box")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin this rank to its GPU's NUMA-local cores")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--n1-value", type=float, default=None, help="garments/s of the N=1 run of the same sweep (tools/run_scale.sh passes it): the line "
                                                                   "then carries scaling_vs_n1 = value / (N x n1)")
    ap.add_argument("--no-latency-b1", action="store_true", help="skip the single-garment latency at the reference's shipped predict configuration (latency_b1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-garments", type=int, default=1)
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 32 if a.workload == "pointnet2" else 16
    return a


class KernelTimer:
    """HIP-event brackets around selected ops.* launches (on torch's current stream = the launch stream of the C ABI)."""

    def __init__(self):
        self.records = []   # (kernel, work, bytes, start_event, end_event)
        self.enabled = False

    def _wrap(self, orig, describe):
        timer = self

        def timed(*a, **kw):
            if not timer.enabled:
                return orig(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **kw)
            e1.record()
            d = describe(out, *a, **kw)
            if d is not None:                       # (None: a launch the roofline does not count, e.g. the gated no-op fp32 twin)
                timer.records.append(d + (e0, e1))
            return out
        return timed

    def install_conv(self):
        from garmentnets_amd import _lib, ops

        def describe(res_, src0, src1, a, d, wp, cout, *rest, **kw):
            B, D, H, W, C0 = src0.shape
            cin = C0 + (0 if src1 is None else src1.shape[-1])
            wbytes = (wp.tensor.numel() * 2.0) if hasattr(wp, "tensor") else wp.numel() * 4.0
            vox = float(B) * D * H * W
            return _lib.load().gn_last_kernel().decode(), 54.0 * cin * cout * vox, (cin + cout) * 4.0 * vox + wbytes

        def describe_up(res_, src1, a1, d1, pack, cout, **kw):       # polyphase partial: 8 coarse taps x 8 parity classes per coarse voxel
            B, Dc, Hc, Wc, C1 = src1.shape
            vox = float(B) * Dc * Hc * Wc
            return "upconv_partial_kernel", 2.0 * 64 * C1 * cout * vox, (C1 + 8 * cout) * 4.0 * vox + pack.tensor.numel() * 2.0

        def describe_ps(res_, src, prep, *rest, **kw):              # the affine-in-weights form: same kernels, per-sample weight packs
            B, D, H, W, cin = src.shape
            vox = float(B) * D * H * W
            return _lib.load().gn_last_kernel().decode(), 54.0 * cin * prep.cout * vox, (cin + prep.cout) * 4.0 * vox + float(prep.pack.numel())

        def describe_w(res_, src, a, d, wp, cout, *rest, **kw):     # Winograd form of the 128-wide kernel: quoted in the DIRECT form's FLOPs (54 Cin Cout per voxel)
            B, D, H, W, cin = src.shape
            vox = float(B) * D * H * W
            return _lib.load().gn_last_kernel().decode(), 54.0 * cin * cout * vox, (cin + cout) * 4.0 * vox + wp.tensor.numel() * 2.0

        ops.conv3d_gcr = self._wrap(ops.conv3d_gcr, describe)
        ops.conv3d_gcr_split_wino = self._wrap(ops.conv3d_gcr_split_wino, describe_w)
        ops.conv3d_gcr_split_persample = self._wrap(ops.conv3d_gcr_split_persample, describe_ps)
        ops.conv3d_gcr_split = self._wrap(ops.conv3d_gcr_split, describe)
        ops.upconv_partial = self._wrap(ops.upconv_partial, describe_up)

    def install_decoder(self):
        """the implicit-decoder launches (networks/conv_implicit_wnf.py:128-149 of the reference): sampler = HBM-bound, MLP = matrix cores.  FLOPs are the
        EXECUTED network's (the folded first layer where the pack is folded), names are what rocprofv3 prints for the same launches"""
        from garmentnets_amd import ops

        def mlp_flops(c0, n, outc, rows):
            return 2.0 * (c0 * n + n * n + n * outc) * rows

        def d_split(res_, xin, pack, out=None, xscale=None):
            M, c0 = xin.shape
            n, oc = pack.hidden, pack.out_channels
            name = f"implicit_decode_split512_kernel<{oc}>" if n == 512 else f"implicit_decode_split_kernel<{oc}, {c0 // 16}, false>"
            return name, mlp_flops(c0, n, oc, M), (c0 + oc) * 4.0 * M + pack.wpack.numel() * 2.0

        def d_lattice(res_, vol_b, Q, pack, out, xscale=None, m0=0, M=None):
            M = Q * Q * Q - m0 if M is None else M
            c0 = vol_b.shape[-1]
            return ("implicit_decode_split_kernel<1, 2, true>", mlp_flops(c0, pack.hidden, pack.out_channels, M),
                    vol_b.numel() * 4.0 * M / float(Q) ** 3 + 4.0 * M + pack.wpack.numel() * 2.0)

        def d_fp32(res_, vol_b, layers, query=None, Q=0, m0=0, M=None, out=None, xin=None, run_if=None):
            if run_if is not None:
                return None                          # gated twin behind a split launch: a no-op unless the device flagged the garment (fp32_twin in the detail file)
            (w1p, _, _, _, n1), (_, _, _, _, n2), (_, _, _, _, oc) = layers
            c0 = xin.shape[1] if xin is not None else vol_b.shape[-1]
            rows = xin.shape[0] if xin is not None else (query.shape[0] if query is not None else M)
            return f"implicit_decode_kernel<{oc}>", 2.0 * (c0 * n1 + n1 * n2 + n2 * oc) * rows, (c0 + oc) * 4.0 * rows + (c0 * n1 + n1 * n2) * 4.0

        def d_samp(res_, vol_b, query=None, Q=0, m0=0, M=None, **k):
            rows = query.shape[0] if query is not None else M
            frac = 1.0 if query is not None or not Q else rows / float(Q) ** 3
            c = vol_b.shape[-1]
            return ("trilinear_kernel" if query is not None else "trilinear_brick_kernel"), 0.0, rows * c * 4.0 + frac * vol_b.numel() * 4.0

        ops.implicit_decode_split = self._wrap(ops.implicit_decode_split, d_split)
        ops.implicit_decode_lattice_split = self._wrap(ops.implicit_decode_lattice_split, d_lattice)
        ops.implicit_decode = self._wrap(ops.implicit_decode, d_fp32)
        ops.trilinear_sample = self._wrap(ops.trilinear_sample, d_samp)

    def install_points(self):
        """PointNet++ operators (config[1]): work = squared-distance evaluations for fps / ball query / kNN, FLOPs for the GEMMs"""
        from garmentnets_amd import ops

        def d_fps(res_, pos, ptr, out_ptr, max_points, m_total, *r, **k):
            B = ptr.numel() - 1
            if k.get("nested_gap") is not None:
                # the cascade's second level (gn_fps_nested): a prefix of the first level's sample wherever that is provably the answer -- the
                # benchmark's clouds have no duplicated points, so NO distance is evaluated; counting the plain sample's evaluations would flatter the rate
                return "fps_kernel (nested: prefix)", 0.0, m_total * 4.0
            return "fps_kernel", float(max_points) * (m_total / max(B, 1)) * B, pos.numel() * 4.0 + m_total * 4.0

        def d_ball(res_, pos, ptr, centre_idx, centre_ptr, r, K=64):
            B = ptr.numel() - 1
            return "ball_query_kernel", float(centre_idx.numel()) * (pos.shape[0] / max(B, 1)), pos.numel() * 4.0 + centre_idx.numel() * (K + 2) * 4.0

        def d_knn(res_, xs, ps, ptr_s, pq, ptr_q, k, **kw):
            B = ptr_s.numel() - 1
            return f"knn_interp_kernel<{k}>", float(pq.shape[0]) * (ps.shape[0] / max(B, 1)), (xs.numel() + pq.shape[0] * xs.shape[1]) * 4.0

        def d_lin(res_, x, w, *r, **kw):
            M, K, N = x.shape[0], (kw.get("K") or x.shape[1]), w.shape[0]
            return "linear_kernel", 2.0 * M * K * N, (M * K + M * N + N * K) * 4.0

        ops.fps = self._wrap(ops.fps, d_fps)
        ops.ball_query = self._wrap(ops.ball_query, d_ball)
        ops.knn_interpolate = self._wrap(ops.knn_interpolate, d_knn)
        ops.linear = self._wrap(ops.linear, d_lin)
        if hasattr(ops, "sa_fused"):
            def d_sa(res_, x, pos, centre_idx, nbr, cnt, pack, **kw):
                M, K = nbr.shape
                return "sa_fused_kernel", 2.0 * M * (K + 1) * pack.macs_per_edge, (M * (K + 1) * (pack.cin + 3) + M * pack.cout) * 4.0
            ops.sa_fused = self._wrap(ops.sa_fused, d_sa)

    def reset(self):
        self.records = []

    def summary(self):
        groups = {}
        for name, work, byts, e0, e1 in self.records:
            g = groups.setdefault(name, dict(work=0.0, bytes=0.0, ms=0.0, n=0))
            g["work"] += work
            g["bytes"] += byts
            g["ms"] += e0.elapsed_time(e1)
            g["n"] += 1
        return groups


GGM_FP64_INSTR_PER_VOXEL = 120.0     # 9 five-tap symmetric correlations (7 fp64 instructions + 5 fp32->fp64 conversions each) + squares, sums, sqrt
GGM_LDS_BYTES_PER_VOXEL = 25.25 * 4.0   # csrc/iso.hip ggm_fused_kernel at its 4 x 8 x 32 tile: staging 3456 + pass 0 444 columns x 16 + pass 1 148 x 48 + pass 2 256 x 28 + 1024 words per 1024 voxels


def fp32_twin_accounting(step):
    """One untimed step with ops.implicit_decode observed.  The f16x2 decoder kernels are followed by a GATED launch of their fp32 twin
    (ops.implicit_decode(run_if=flag)): a no-op unless gn_decoder_input_scale marked the garment unsafe for fp16 planes (csrc/decode_split.hip).
    -> per decoder (by output width): gated launches per step, how many of them ran, how many distinct garments were flagged -- read from the
    device flags after the step, outside every timed region; plus fp32 decoder launches that were not gated at all."""
    from garmentnets_amd import ops
    calls = []
    orig = ops.implicit_decode

    def spy(vol_b, layers, *a, **kw):
        calls.append((kw.get("run_if"), int(layers[2][4])))
        return orig(vol_b, layers, *a, **kw)
    ops.implicit_decode = spy
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.implicit_decode = orig
    per, ungated = {}, 0
    for flag, outc in calls:
        if flag is None:
            ungated += 1
            continue
        d = per.setdefault(f"out{outc}", {"gated_launches_per_step": 0, "ran": 0, "_garments": set()})
        d["gated_launches_per_step"] += 1
        if float(flag.reshape(-1)[0]) != 0.0:
            d["ran"] += 1
            d["_garments"].add(flag.data_ptr())
    for d in per.values():
        d["garments_on_the_fp32_twin"] = len(d.pop("_garments"))
    return {"decoders": per, "ungated_fp32_decoder_launches_per_step": ungated,
            "what": "gated launches of the fp32 decoder kernel behind every f16x2 decoder launch (implicit_decode_kernel<OUT>): `ran` of them did work "
                    "(the device-side `unsafe` flag of gn_decoder_input_scale), the others return after reading the flag.  In a kernel trace their "
                    "duration is queue residency: beside a batch in flight on another stream a no-op launch waits for a free CU like any other"}


def latency_b1(args, dev, runs=21):
    """ONE garment through predict_batch at the reference's SHIPPED predict configuration (config/predict_default.yaml:5,43 + predict.py:62: batch 1,
    32^3 feature volume with max reduction, 128^3 WNF lattice -- what `python -m garmentnets_amd.predict` defaults to), library-default arithmetic:
    median wall time of `runs` synchronised runs, and the stage split of one more run (HIP events).  Every throughput figure of the line is a
    batch of 16 or 8; here the serial farthest-point sampling (one workgroup per garment) and the launch gaps are what is left."""
    from garmentnets_amd.arith import Arith
    from garmentnets_amd.batch import Batch
    from garmentnets_amd.common import marching_cubes_util as mcu
    from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
    from garmentnets_amd.predict import _iso_capacity, predict_batch
    hp, sd, shard, _ = bench_inputs(1, args.points, 32, "max", args.input)
    model = ConvImplicitWNFPipeline(**hp)
    model.load_state_dict(sd)
    model = model.to(dev).eval().requires_grad_(False)
    model.arith = Arith.named("f16x2", "f16x2")
    data = Batch(sizes=shard.sizes, x=shard.x, pos=shard.pos, batch=shard.batch).to(dev)
    Q = 128
    run = lambda: predict_batch(model, data, volume_size=Q, iso_surface_level=0.5, gradient_sigma=0.5, gradient_direction="ascent")
    res = None
    for _ in range(3):
        res = run()
    torch.cuda.synchronize()
    if any(bool(torch.isnan(r["verts"]).any()) for r in res):
        return {"skipped": "the synthetic checkpoint's WNF does not straddle the 0.5 level at G = 32"}
    ts = []
    for _ in range(runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    with torch.no_grad():
        ev[0].record()
        p2 = model.pointnet2_forward(data, prefetch_volume=True)
        ev[1].record()
        u3 = model.unet3d_forward(p2)
        ev[2].record()
        wnf = model.volume_lattice_forward(u3, Q)["pred_volume"]
        ev[3].record()
        job = mcu.IsoBatchJob(Q, 0.5, 0.5, "ascent", cap_v=_iso_capacity(model, Q))
        job.enqueue(wnf)
        job.finish()
        q_all = job.padded_queries()
        if q_all is not None:
            model.surface_decoder_forward(u3, q_all)
        ev[4].record()
        torch.cuda.synchronize()
    names = ("pointnet2_forward", "unet3d_forward (gridding + UNet)", "volume_lattice_forward (sampler + decoder)", "GGM + MC33 + surface decode")
    return {"ms_median": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1], "runs": runs, "garments_per_s": 1e3 / ts[len(ts) // 2],
            "config": "batch 1, 6000-pt cloud, 32^3 feature volume (max), 128^3 WNF + GGM + MC33 + surface decode: config/predict_default.yaml:5,43, predict.py:62",
            "verts": int(res[0]["verts"].shape[0]), "stages_ms": {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)},
            "what": "wall time of predict_batch for ONE garment, inputs resident, results left on the device, host synchronised before and after each run"}


def measured_roofs():
    """fp64 vector rate and LDS read rate measured on THIS box by tools/dev/roof_burn (built by __graft_entry__.build()): the ceilings of
    the members whose arithmetic is fp64 by contract (scipy / scikit-image bit-exactness).  -> dict or None"""
    import subprocess
    exe = os.path.join(REPO, "tools", "dev", "_build", "roof_burn")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None


class HbmMembers:
    """HIP-event brackets around the HBM-bound members of the path (zero-fill, scatter, max-pool, sampler, GGM, min/max,
    MC33) during ONE untimed step; bytes = ALGORITHMIC bytes of the call (each input / output once), fraction of the 8 TB/s HBM3E peak.
    mc33 is one C-ABI call that launches classify / scan / vertices+attributes / faces: its stage split comes from
    gn_mc33_batch_profiled (events between the stages inside the call)."""

    def __init__(self):
        self.records, self.saved = [], {}

    def _wrap(self, name, nbytes):
        from garmentnets_amd import ops
        orig = getattr(ops, name)
        self.saved[name] = orig
        me = self

        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **kw)
            e1.record()
            me.records.append((name, float(nbytes(out, *a, **kw)), e0, e1))
            return out
        setattr(ops, name, timed)

    def install(self):
        nb = lambda t: t.numel() * t.element_size()
        self._wrap("zeroed_volume", lambda res, *a, **k: nb(res[0]) + nb(res[1]))
        self._wrap("grid_scatter", lambda res, src, flat, *a, **k: 2 * nb(src) + nb(flat))
        self._wrap("maxpool3d_2", lambda res, x, *a, **k: nb(x) + nb(res[0] if isinstance(res, tuple) else res))

        def samp(res, vol_b, query=None, Q=0, m0=0, M=None, **k):
            rows = query.shape[0] if query is not None else M
            frac = 1.0 if query is not None or not Q else rows / float(Q) ** 3
            return rows * vol_b.shape[-1] * 4 + frac * nb(vol_b)
        self._wrap("trilinear_sample", samp)
        self._wrap("ggm3d_batch", lambda res, vols, *a, **k: 2 * nb(vols))
        self._wrap("ggm3d_batch_range", lambda res, vols, *a, **k: 2 * nb(vols))      # (round 6: the predict path's form -- the volumes' (min, max) ride along)
        self._wrap("minmax_batch", lambda res, vols, *a, **k: nb(vols))
        self._wrap("mc33_batch", lambda res, vols, *a, **k: nb(vols))          # (+ V * 28 + F * 12 of mesh output: < 1 % of the volume bytes)
        self._wrap("gather_nn_batch", lambda res, vols, verts, *a, **k: nb(verts) + nb(res))

    def uninstall(self):
        from garmentnets_amd import ops
        for k, v in self.saved.items():
            setattr(ops, k, v)
        self.saved = {}

    def reset(self):
        self.records = []

    def summary(self, wnf_all, level):
        from garmentnets_amd import ops
        groups = {}
        for name, byts, e0, e1 in self.records:
            g = groups.setdefault(name, dict(bytes=0.0, ms=0.0, calls=0))
            g["bytes"] += byts
            g["ms"] += e0.elapsed_time(e1)
            g["calls"] += 1
        if "ggm3d_batch_range" in groups:            # one name for the GGM launch across rounds
            groups["ggm3d_batch"] = groups.pop("ggm3d_batch_range")
        out = {}
        for name, g in groups.items():
            gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0
            out[name] = {"calls": g["calls"], "ms": g["ms"], "algorithmic_bytes": g["bytes"], "GBs": gbs, "frac_of_8TBs": gbs / PEAK_HBM_GBS}
        if hasattr(ops, "mc33_batch_profiled") and wnf_all is not None:
            st = ops.mc33_batch_profiled(wnf_all, level)
            stages = {"mesh_vertices_per_batch": st.pop("_mesh")[1]}
            for k, (ms, byts) in st.items():
                gbs = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                stages[k] = {"ms": ms, "algorithmic_bytes": byts, "GBs": gbs, "frac_of_8TBs": gbs / PEAK_HBM_GBS}
            out["mc33_stages"] = stages
            # the same two operators on SURVEY.md 8d's analytic shell volume (a smooth, garment-like closed surface: ~40 k vertices per 128^3
            # volume instead of the ~500 k of the random-weight WNF): one batch of identical volumes, timed by HIP events
            from garmentnets_amd import synthetic as S
            B, Q = wnf_all.shape[0], wnf_all.shape[-1]
            shell = torch.from_numpy(S.shell_volume(Q)).to(wnf_all.device).expand(B, Q, Q, Q).contiguous()
            ops.ggm3d_batch(shell, 0.5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.ggm3d_batch(shell, 0.5)
            e1.record()
            st2 = ops.mc33_batch_profiled(shell, 0.5)
            ms_g = e0.elapsed_time(e1)
            byts = 2.0 * shell.numel() * 4
            sh = {"ggm3d_batch": {"ms": ms_g, "algorithmic_bytes": byts, "GBs": byts / (ms_g * 1e-3) / 1e9, "frac_of_8TBs": byts / (ms_g * 1e-3) / 1e9 / PEAK_HBM_GBS},
                  "mesh_vertices_per_batch": st2.pop("_mesh")[1]}
            for k, (ms, b_) in st2.items():
                gbs = b_ / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                sh[k] = {"ms": ms, "algorithmic_bytes": b_, "GBs": gbs, "frac_of_8TBs": gbs / PEAK_HBM_GBS}
            out["iso_on_shell_volume"] = sh
            # the GGM's two accumulation widths on the step's own volumes (Arith.ggm_fp32; the default is scipy's fp64 arithmetic bit for bit)
            ab = {}
            for bits in (64, 32):
                best, g_ = None, None
                for _ in range(4):                                   # (the first calls of a variant pay its allocations: fastest of the later ones)
                    g_ = None
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g_ = ops.ggm3d_batch_range(wnf_all, 0.5, bits)[0]
                    e1.record()
                    torch.cuda.synchronize()
                    best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
                ab["fp%d" % bits] = (best, g_)
            ref = ab["fp64"][1].double()
            out["ggm_accumulation"] = {"fp64_ms": ab["fp64"][0], "fp32_ms": ab["fp32"][0],
                                       "fp32_max_abs_diff_vs_fp64": float((ab["fp32"][1].double() - ref).abs().max()),
                                       "ggm_max": float(ref.max()), "note": "with the (min, max) record riding along in both"}
        roofs = measured_roofs()
        if roofs is not None:
            out["roofs"] = dict(roofs, public_spec={"fp64_vector_tflops": 78.6, "lds_TBs": "MI355X_MICROARCH.md: ds_read_b128 256 B/clk/CU = 157 TB/s at 2.4 GHz, ~150 measured with every CU streaming"},
                                note="measured on this box right before this line was printed (tools/dev/roof_burn.hip)")
            instr_rate = roofs["fp64_fma_tflops"] * 1e12 / 2.0          # fp64 vector instructions per second (an FMA counts 2 FLOP)

            def ggm_roofs(entry, voxels):
                sec = entry["ms"] * 1e-3
                entry["fp64_instr_per_voxel"] = GGM_FP64_INSTR_PER_VOXEL
                entry["frac_of_fp64_rate"] = GGM_FP64_INSTR_PER_VOXEL * voxels / sec / instr_rate
                entry["lds_bytes_per_voxel"] = GGM_LDS_BYTES_PER_VOXEL
                entry["frac_of_lds_rate"] = GGM_LDS_BYTES_PER_VOXEL * voxels / sec / (roofs["lds_read_TBs"] * 1e12)
            if "ggm3d_batch" in out and wnf_all is not None:
                ggm_roofs(out["ggm3d_batch"], float(wnf_all.numel()) * out["ggm3d_batch"]["calls"])
            if "iso_on_shell_volume" in out and wnf_all is not None:
                ggm_roofs(out["iso_on_shell_volume"]["ggm3d_batch"], float(wnf_all.numel()))
            for grp in (out.get("mc33_stages"), out.get("iso_on_shell_volume")):
                if grp and "mesh_vertices_per_batch" in grp and grp["mesh_vertices_per_batch"]:
                    nv = float(grp["mesh_vertices_per_batch"])
                    for k, v in grp.items():
                        if isinstance(v, dict) and k.startswith("mc_") and "ms" in v:
                            v["ns_per_vertex"] = v["ms"] * 1e6 / nv
        out["note"] = ("one untimed step; bytes = algorithmic (each input / output of the call once); scatter is atomics / latency-bound by nature "
                       "(6000 points per garment), zero-fill runs on a side stream beside farthest-point sampling; GGM is ONE fused launch at the algorithmic minimum of HBM bytes (one read, one write); its three "
                       "fractions (frac_of_8TBs, frac_of_fp64_rate, frac_of_lds_rate against the ceilings measured in `roofs`) are ALL small: it is bound by "
                       "none of the three but by latency -- 9 dependent five-tap fp64 chains per voxel (scipy's arithmetic, bit for bit), four barrier-separated phases; "
                       "round 6: 4 x 8 x 32 tiles (28 KB of LDS: 5 waves per SIMD instead of 2) took it from 0.72 to 0.42 ms, and the volumes' (min, max) ride on its staging pass; the MC33 "
                       "stages do per-cell fp64 case analysis and per-vertex fp64 gathers: their time follows the number of surface cells, not the bytes")
        return out


def pmc_child(args, model, data, step):
    """what the rocprofv3 child passes run: one warm-up and one measured step of the headline arithmetic, nothing else"""
    step()
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()


def measure_traffic(args):
    """HBM bytes per launch of every kernel of one headline step, measured on THIS box in THIS run: two rocprofv3 child passes of
    `bench.py --pmc-child` (same workload arguments; --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md's HBM section prescribes; gfx950 FETCH_SIZE correction per kernel: tools/pmc_summary.py).
    -> ({kernel: {...bytes per launch...}}, note) or (None, why not)"""
    import shutil
    import subprocess
    import tempfile
    torch.cuda.empty_cache()                 # the child allocates the same working set on this GPU
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import pmc_summary
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--batch", str(args.batch), "--points", str(args.points),
             "--grid", str(args.grid), "--reduce", args.reduce, "--volume-size", str(args.volume_size), "--input", args.input,
             "--conv-mode", args.conv_mode, "--decode-mode", args.decode_mode, "--winograd", args.winograd, "--winograd32", args.winograd32]
    env = dict(os.environ, TMPDIR="/tmp", GARMENTNETS_PREFETCH_ZERO="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="gn_pmc_", dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", os.path.join(tmp, counter), "-o", "pmc", "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} child pass timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child pass failed (rc {r.returncode}): " + r.stdout.decode(errors="replace")[-300:]
        try:
            summ = pmc_summary.summarise(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"))
        except Exception as e:      # noqa: BLE001
            return None, f"could not read the counter CSVs: {e!r}"
    return summ["kernels"], (f"this run, this box: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of one headline step "
                             f"({time.time() - t0:.0f} s); FETCH_SIZE x per-kernel gfx950 factor (tools/pmc_summary.py), WRITE_SIZE as reported")


def is_conv_group(name):
    return name.startswith("conv3d") or name.startswith("upconv")


def is_decoder_group(name):
    return name.startswith("implicit_decode")


def pick_roofline(args, groups, conv_mode, decode_mode, hw=None, traffic=None):
    """the roofline of the kernel with the MOST TIME in it during the timed steps, over every bracketed group (convolutions, decoder MLPs, sampler,
    PointNet++ operators) -- at Q = 256 or G = 32 that is the decoder MLP, not a convolution.  `by_group_ms` lists the contenders."""
    key = max(groups, key=lambda k: groups[k]["ms"])
    if is_conv_group(key):
        rl = conv_roofline(args, {k: v for k, v in groups.items() if is_conv_group(k)}, conv_mode, hw, traffic)
    elif is_decoder_group(key):
        rl = decoder_roofline(groups, key, decode_mode, hw, traffic)
    else:
        rl = points_roofline({k: v for k, v in groups.items() if not (is_conv_group(k) or is_decoder_group(k))})
    total = sum(v["ms"] for v in groups.values())
    rl["share_of_bracketed_ms"] = groups[rl["kernel"]]["ms"] / total if total > 0 else None
    rl["by_group_ms"] = {k: round(v["ms"], 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}
    return rl


def decoder_roofline(groups, key, decode_mode, hw=None, traffic=None):
    """the implicit decoder's MLP kernel (csrc/decode_split.hip, csrc/decode.hip): algorithmic FLOPs of the executed network / HIP-event time"""
    g = groups[key]
    sec = g["ms"] * 1e-3
    achieved = g["work"] / sec / 1e12
    split = "split" in key
    n = SPLIT_PRODUCTS["f16x2"] if split else 1
    peak = PEAK_16BIT_MFMA_TFLOPS / n if split else PEAK_FP32_MFMA_TFLOPS
    tr, src, tr_detail = None, None, None
    if traffic is not None:
        table, src = traffic
        if table is not None and key in table:
            tr, tr_detail = table[key]["hbm_bytes"], table[key]
    out = {"bound": "mfma", "kernel": key, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
           "peak_note": (f"16-bit MFMA dense peak {PEAK_16BIT_MFMA_TFLOPS:.0f} / {n} executed f16 products per fp32 product (hi*hi + hi*lo + lo*hi); "
                         f"executed {achieved * n:.0f} TFLOP/s" if split else "fp32 MFMA dense peak"),
           "traffic": tr, "traffic_source": src, "traffic_detail": tr_detail, "launches": g["n"], "avg_launch_ms": g["ms"] / g["n"],
           "flops_per_launch": g["work"] / g["n"], "algorithmic_bytes_per_launch": g["bytes"] / g["n"],
           "hbm_frac_of_8TBs": g["bytes"] / sec / 1e9 / PEAK_HBM_GBS,
           "all_decoder_instances": {k: {"launches": v["n"], "ms": v["ms"], "tflops": v["work"] / (v["ms"] * 1e-3) / 1e12}
                                     for k, v in groups.items() if is_decoder_group(k)}}
    if hw is not None:
        sclk = hw.get("sclk_mhz")
        out.update(sclk_mhz=sclk, socket_power_w=hw.get("socket_power_w"), power_cap_w=hw.get("power_cap_w"),
                   frac_at_2400mhz=(out["frac"] * 2400.0 / sclk) if sclk else None, hwmon=hw)
    return out


def conv_roofline(args, groups, conv_mode, hw=None, traffic=None):
    """groups: the convolution groups only; hw: HwmonSampler.summary() of the pass; traffic: (measure_traffic's kernel table, note) of this run"""
    key = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[key]
    achieved = g["work"] / (g["ms"] * 1e-3) / 1e12          # algorithmic (fp32) FLOPs: 54*Cin*Cout per voxel
    if conv_mode == "fp32":
        peak, peak_note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense peak"
    else:
        n = SPLIT_PRODUCTS[conv_mode]
        wino = "wino" in key
        if wino:            # Winograd F(2,3) along x: 36 of the direct form's 54 tap products per output pair (csrc/unet_wino.hip)
            n = n * 36.0 / 54.0
        peak = PEAK_16BIT_MFMA_TFLOPS / n
        peak_note = (f"16-bit MFMA dense peak {PEAK_16BIT_MFMA_TFLOPS:.0f} / {n:g} EXECUTED matrix-core products per algorithmic fp32 product "
                     f"({conv_mode}" + (" x 36/54: Winograd F(2,3) along x; `achieved` counts the DIRECT form's 54*Cin*Cout FLOPs per voxel, so frac is the "
                                        "matrix pipe's utilisation, and achieved / (2500 / 3) is the speed in the direct form's terms" if wino else "") +
                     f"); executed {achieved * n:.0f} TFLOP/s; the fp32-MFMA peak is {PEAK_FP32_MFMA_TFLOPS}")
    tr, src, tr_detail = None, None, None
    if traffic is not None:
        table, src = traffic
        if table is not None and key in table:
            tr, tr_detail = table[key]["hbm_bytes"], table[key]
    out = {"bound": "mfma", "kernel": key, "kernel_label": "reported by the C ABI (gn_last_kernel) after each launch",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
            "traffic": tr, "traffic_source": src, "traffic_detail": tr_detail,
            "launches": g["n"], "avg_launch_ms": g["ms"] / g["n"], "flops_per_launch": g["work"] / g["n"],
            "algorithmic_bytes_per_launch": g["bytes"] / g["n"],
            "hbm_frac_of_8TBs": g["bytes"] / (g["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "frac_of_direct_form_peak": achieved / (PEAK_16BIT_MFMA_TFLOPS / SPLIT_PRODUCTS[conv_mode]) if conv_mode != "fp32" else None,
            "all_conv_instances": {k: {"launches": v["n"], "ms": v["ms"], "tflops": v["work"] / (v["ms"] * 1e-3) / 1e12,
                                       "frac": v["work"] / (v["ms"] * 1e-3) / 1e12 / (
                                           PEAK_FP32_MFMA_TFLOPS if conv_mode == "fp32" else PEAK_16BIT_MFMA_TFLOPS / (SPLIT_PRODUCTS[conv_mode] * (36.0 / 54.0 if "wino" in k else 1.0)))}
                                   for k, v in groups.items()}}
    if hw is not None:
        sclk = hw.get("sclk_mhz")
        out.update(sclk_mhz=sclk, socket_power_w=hw.get("socket_power_w"), power_cap_w=hw.get("power_cap_w"),
                   frac_at_2400mhz=(out["frac"] * 2400.0 / sclk) if sclk else None,
                   clock_note="sclk / socket power: mean of the GPU's hwmon nodes sampled every 50 ms during this pass (all kernels of the step, "
                              "not the dominant one alone); the peak assumes 2400 MHz, frac_at_2400mhz = frac x 2400 / sclk is the matrix-core "
                              "issue rate the kernel sustains per clock; busy counters: profiles/", hwmon=hw)
    return out


def points_roofline(groups):
    """config[1]: per-operator rates in the units SURVEY.md 8d names (distance evaluations / s, FPS steps / s, GEMM TFLOP/s); the
    `roofline` object proper is the dominant operator by time"""
    key = max(groups, key=lambda k: groups[k]["ms"])
    per = {}
    for k, v in groups.items():
        sec = v["ms"] * 1e-3
        per[k] = {"launches": v["n"], "ms": v["ms"], "work_per_s": v["work"] / sec, "algorithmic_GBs": v["bytes"] / sec / 1e9,
                  "work_unit": "FLOP" if ("linear" in k or "sa_fused" in k) else "squared-distance evaluations"}
    g = groups[key]
    sec = g["ms"] * 1e-3
    if "linear" in key or "sa_fused" in key:
        rl = {"bound": "mfma", "kernel": key, "achieved": g["work"] / sec / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s"}
    else:   # fps / ball query / kNN: latency- or ALU-bound scans; against the HBM roofline their compulsory traffic is negligible -- say so
        rl = {"bound": "hbm", "kernel": key, "achieved": g["bytes"] / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "note": "serial / ALU-bound scan (fps: one dependent arg-max per sample); the algorithmic bytes are tiny by nature, the operator "
                      "rate is in per_operator.work_per_s"}
    rl["frac"] = rl["achieved"] / rl["peak"]
    rl.update(traffic=None, launches=g["n"], avg_launch_ms=g["ms"] / g["n"], per_operator=per)
    return rl


COMPACT_LIMIT = 6144            # bytes: the driver recovers the line from the tail of stdout (round 5's 21.6 KB line did not survive it)
COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "value_is", "with_host_io", "two_in_flight_with_host_io", "literal_affine", "strict_fp32",
                "occupancy_aware", "latency_b1_ms", "stages_ms", "oracle_check", "validation_ok", "rccl_ranks_seen", "dist_backend", "per_rank",
                "scaling_vs_n1", "detail")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches",
                 "share_of_bracketed_ms", "sclk_mhz", "socket_power_w", "ppt_residency")


def _sig(x, n=5):
    """numbers to n significant digits (the line is read by a parser and by people; the detail file keeps every digit)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    return float(f"{float(x):.{n}g}")


def _short_roofline(rl):
    if rl is None:
        return None
    hw = rl.get("hwmon") or {}
    thr = hw.get("throttle") or rl.get("throttle") or {}
    d = {k: _sig(rl.get(k)) for k in ROOFLINE_KEYS if k != "ppt_residency"}
    d["ppt_residency"] = _sig(thr.get("ppt_power_limit_residency"))
    return d


def compact_line(full, detail_path=None):
    """The ONE line rank 0 prints on stdout: the contract's keys first and only scalars / short strings, <= COMPACT_LIMIT bytes (tests/test_bench_line.py);
    everything else of `full` (hwmon traces, throttle residencies, hbm_members, per-kernel tables, the prose) goes to the detail file and to stderr."""
    cfg = full.get("config") or {}
    wl = cfg.get("workload") or ""
    short_cfg = {"workload": wl.split("; input:")[0], "input": cfg.get("input"), "batch_per_gpu": cfg.get("batch_per_gpu"), "global_batch": cfg.get("global_batch"),
                 "points": cfg.get("points"), "grid": cfg.get("grid"), "reduce": cfg.get("reduce"), "volume_size": cfg.get("volume_size"),
                 "mesh_verts_per_garment": _sig(cfg.get("mesh_verts_per_garment")), "parallelism": (cfg.get("parallelism") or "").split(" (")[0]}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    out["value"], out["ms_per_step"] = _sig(out["value"], 6), _sig(out["ms_per_step"], 6)
    out["dtype"] = full.get("dtype_short") or ((full.get("dtype") or "").split(" on the 16-bit")[0] + (", fp32 accumulation)" if " on the 16-bit" in (full.get("dtype") or "") else ""))
    out["data"] = full.get("data")
    out["config"] = short_cfg
    out["roofline"] = _short_roofline(full.get("roofline"))
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = None if cb is None else {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                                   "sample": cb["sample"].split(", oracle/")[0][:160]}
    out["value_is"] = "inputs resident in HBM, results left on the device (bench contract); with_host_io = SURVEY 8d: H2D clouds + D2H meshes inside the timed region"
    sub = lambda name: full.get(name) or {}
    pair = lambda name: ({"value": _sig(sub(name).get("value")), "ms_per_step": _sig(sub(name).get("ms_per_step"))} if full.get(name) else None)
    out["with_host_io"] = pair("with_host_io")
    out["two_in_flight_with_host_io"] = pair("two_in_flight_with_host_io")
    for name in ("literal_affine", "strict_fp32"):
        if full.get(name):
            rl = sub(name).get("roofline") or {}
            out[name] = {"value": _sig(sub(name).get("value")), "kernel": rl.get("kernel"), "frac": _sig(rl.get("frac")), "achieved": _sig(rl.get("achieved")),
                         "peak": _sig(rl.get("peak")), "sclk_mhz": _sig(rl.get("sclk_mhz"))}
        else:
            out[name] = None
    out["occupancy_aware"] = pair("occupancy_aware")
    out["latency_b1_ms"] = _sig(sub("latency_b1").get("ms_median")) if full.get("latency_b1") else None
    st = full.get("stages_ms")
    out["stages_ms"] = None if st is None else {k.split(" (")[0]: _sig(v, 4) for k, v in st.items()}
    oc = full.get("oracle_check")
    out["oracle_check"] = None if oc is None else {k: _sig(oc.get(k)) for k in ("ok", "wnf_max_abs_err", "nocs_bins_equal", "occupied_cells_equal", "features_max_abs_err")
                                                   if k in oc}
    out["validation_ok"] = (full.get("validation") or {}).get("ok")
    out["rccl_ranks_seen"] = full.get("rccl_ranks_seen")
    out["dist_backend"] = full.get("dist_backend")
    pr = full.get("per_rank") or {}
    out["per_rank"] = {"slowest_over_fastest": _sig(pr.get("slowest_over_fastest")),
                       "garments_per_s": [_sig(v, 4) for v in (pr.get("garments_per_s") or [])][:16]}
    out["scaling_vs_n1"] = _sig(full.get("scaling_vs_n1"))
    out["detail"] = detail_path
    assert tuple(out) == COMPACT_KEYS, tuple(out)
    return out


def emit(full):
    """detail -> gpurun_out/bench_detail.json (+ stderr), the compact line -> stdout, LAST"""
    detail_path = None
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        detail_path = os.path.join("gpurun_out", "bench_detail.json")
        with open(os.path.join(REPO, detail_path), "w") as f:
            json.dump(full, f)
    except OSError:
        detail_path = None
    print("[bench detail] " + json.dumps(full), file=sys.stderr, flush=True)
    text = json.dumps(compact_line(full, detail_path))
    assert len(text) <= COMPACT_LIMIT, len(text)
    sys.stdout.flush()
    print(text, flush=True)


def bench_inputs(batch, points, grid, reduce, input_kind, rank=0, world=1):
    """the benchmark's model inputs: (hparams, state dict, this rank's shard of the seeded global batch on the host, (lo, hi)).
    tests/test_gpu_fullsize.py::test_bench_batch_against_oracle calls this to check THE batch the number is quoted on."""
    from garmentnets_amd import parallel, synthetic as S
    planted = input_kind in ("planted", "noisy_wnf")
    hp = S.default_hparams(grid=grid, reduce_method=reduce)
    sd = S.synthetic_state_dict(hp, 0, planted_nocs=planted, planted_wnf=input_kind == "planted")
    shard, span = parallel.shard_batch(batch * world, points, CLOUD_SEED, rank, world, colour="position" if planted else "uniform")
    return hp, sd, shard, span


def cpu_baseline(args, hp, sd, shard, probe):
    """The oracle (a torch-CPU port of the reference path: 'port') on this host's cores, timed on garment 0 of the benchmark batch
    (PointConv's self-loop quirk links centre i to point i of the batch: for garment 0 that is the garment itself, so its result alone
    IS its result in the batch).  -> (cpu_baseline dict, oracle_check dict or None): the oracle's result for that garment against slot
    0 of the HIP path's (probe: tensors kept from an untimed step)"""
    from garmentnets_amd import synthetic as S
    from oracle import pipeline as P
    ncpu = os.cpu_count() or 1
    # pick the thread count the host actually runs this path fastest with (all cores is NOT it: torch-CPU conv3d collapses
    # under oversubscription -- 256 threads were 27x slower than 32 on the MI355X host); probe = UNet on a 32^3 volume
    probe_hp = S.default_hparams(grid=32)
    probe_sd = S.synthetic_state_dict(probe_hp, 0)
    xprobe = torch.randn(1, 128, 32, 32, 32)
    best, cores = None, 1
    for nt in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            P.unet3d(probe_sd, probe_hp["unet3d_params"], xprobe[:, :, :8, :8, :8])
            t0 = time.time()
            P.unet3d(probe_sd, probe_hp["unet3d_params"], xprobe)
            dtp = time.time() - t0
        if best is None or dtp < best:
            best, cores = dtp, nt
    torch.set_num_threads(cores)
    n = args.cpu_baseline_garments
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    npts = args.points
    x, pos, batch = shard.x[:n * npts].clone(), shard.pos[:n * npts].clone(), shard.batch[:n * npts].clone()
    check = None
    if args.workload == "pointnet2":
        n = max(n, 4)
        x, pos, batch = shard.x[:n * npts].clone(), shard.pos[:n * npts].clone(), shard.batch[:n * npts].clone()
        with torch.no_grad():
            P.pointnet2_forward(sd_cpu, hp, x[:npts], pos[:npts], batch[:npts])
            t0 = time.time()
            ref = P.pointnet2_forward(sd_cpu, hp, x, pos, batch)
            dt = time.time() - t0
        what = f"garments 0..{n - 1} of the benchmark batch, PointNet2NOCS forward + NOCS post-processing (N={npts})"
        if probe is not None:
            check = {"garment": 0, "nocs_bins_equal": bool(torch.equal(ref["nocs_data"]["nocs_bin_idx"][:npts], probe["bins0"])),
                     "features_max_abs_err": float((ref["per_point_features"][:npts] - probe["feat0"]).abs().max())}
            check["ok"] = check["nocs_bins_equal"] and check["features_max_abs_err"] <= 1e-4
    else:
        level = probe["level0"] if probe is not None else 0.5
        t0 = time.time()
        ref = P.predict(sd_cpu, hp, x, pos, batch, Q=args.volume_size, level=level, sigma=0.5, auto_level=False)
        dt = time.time() - t0
        what = f"garment(s) 0..{n - 1} of the benchmark batch (N={npts}, G={args.grid} {args.reduce}, Q={args.volume_size})"
        if probe is not None:
            g0 = ref["garments"][0]
            occ_ref = (ref["in_feature_volume"][0] != 0).any(dim=0)
            wnf_err = float(np.abs(g0["wnf_volume"] - probe["wnf0"].numpy()).max())
            check = {"garment": 0, "what": "oracle/pipeline.py on garment 0 of the benchmark batch vs slot 0 of the HIP path (the arithmetic of the headline pass)",
                     "nocs_bins_equal": bool(torch.equal(ref["pointnet2_result"]["nocs_data"]["nocs_bin_idx"][:npts], probe["bins0"])),
                     "occupied_cells": int(occ_ref.sum()), "occupied_cells_equal": bool(torch.equal(occ_ref, probe["occ0"])),
                     "wnf_max_abs_err": wnf_err, "wnf_tolerance": 1e-4, "iso_level": level,
                     "oracle_mesh": {"verts": int(len(g0["verts"])) if "verts" in g0 else None, "faces": int(len(g0["faces"])) if "faces" in g0 else None},
                     "hip_mesh": {"verts": probe["verts0"], "faces": probe["faces0"]}}
            check["ok"] = check["nocs_bins_equal"] and check["occupied_cells_equal"] and wnf_err <= 1e-4
    base = {"value": n / dt, "unit": "garments/s", "cores": cores, "kind": "port",
            "sample": f"{what}, oracle/pipeline.py on torch-CPU fp32 with {cores} of {ncpu} hardware threads (fastest of a short sweep; fps / ball "
                      f"query / kNN / GGM / marching cubes single-threaded C as in the reference), {dt:.1f} s"}
    return base, check


def validate(model, args, dev, auto_level):
    """untimed: `batch` IDENTICAL garments with the PointConv self-loop quirk off (it links centre i to point i of the whole batch, so a
    garment's result legitimately depends on its slot otherwise) -- slot 0 and slot B-1 must agree"""
    from garmentnets_amd import synthetic as S
    from garmentnets_amd.batch import Batch
    from garmentnets_amd.predict import predict_batch
    B, n = args.batch, args.points
    pn = model.pointnet2_nocs
    saved = (pn.sa1_module.conv.add_self_loops, pn.sa2_module.conv.add_self_loops)
    pn.sa1_module.conv.add_self_loops = pn.sa2_module.conv.add_self_loops = False
    try:
        x, pos, _ = S.synthetic_cloud(1, n, seed=4242, colour="position" if args.input != "collapsed" else "uniform")
        data = Batch(sizes=[n] * B, x=x.repeat(B, 1), pos=pos.repeat(B, 1), batch=torch.arange(B).repeat_interleave(n)).to(dev)
        if args.workload == "pointnet2":
            with torch.no_grad():
                p2 = model.pointnet2_forward(data)
            lg = p2["per_point_logits"]
            same = bool(torch.equal(lg[:n], lg[(B - 1) * n:])) and bool(torch.equal(p2["global_feature"][0], p2["global_feature"][B - 1]))
            out = {"identical_garments": B, "logits_and_global_feature_slot0_eq_slotlast": same}
            ok = same
        else:
            res = predict_batch(model, data, volume_size=args.volume_size, iso_surface_level=0.5, gradient_sigma=0.5, auto_level=auto_level)
            w0, w1 = res[0]["wnf_volume"], res[B - 1]["wnf_volume"]
            spread = float((w0 - w1).abs().max())
            faces_equal = res[0]["faces"].shape == res[B - 1]["faces"].shape and bool(torch.equal(res[0]["faces"], res[B - 1]["faces"]))
            dv = float((res[0]["verts"] - res[B - 1]["verts"]).abs().max()) if faces_equal else None
            out = {"identical_garments": B, "wnf_max_abs_diff_slot0_vs_slotlast": spread, "faces_equal": faces_equal, "verts_max_abs_diff": dv,
                   "verts": int(res[0]["verts"].shape[0]), "wnf_checksum_slot0": float(w0.double().sum()), "wnf_checksum_slotlast": float(w1.double().sum())}
            ok = spread <= 1e-5 and bool(torch.isfinite(w1).all()) and (faces_equal or spread > 0)
        out["ok"] = bool(ok)
        return out
    finally:
        pn.sa1_module.conv.add_self_loops, pn.sa2_module.conv.add_self_loops = saved


def main():
    args = parse()
    from garmentnets_amd import parallel
    rank, local_rank, world = parallel.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # (GARMENTNETS_DIST_BACKEND=gloo: the N > 1 path on a box with fewer GPUs than ranks -- ranks share devices; tests/test_gpu_api.py)
    backend = os.environ.get("GARMENTNETS_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    parallel.init(backend=backend, device=dev)     # "nccl" is RCCL on ROCm; no-op for one process
    # one host process per GPU: keep each rank's host side (Python, mesh slicing, pinned D2H staging) on its GPU's NUMA-local cores
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    full_mask = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if args.no_affinity or world == 1:
        affinity = {"pinned": False, "why_not": "--no-affinity" if args.no_affinity else "single rank: nothing shares the host"}
    else:
        affinity = parallel.pin_rank(local_rank, local_world, dev_index)
    metrics_dev = dev if backend == "nccl" else "cpu"

    from garmentnets_amd import ops
    from garmentnets_amd.arith import Arith
    from garmentnets_amd.batch import Batch
    from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
    from garmentnets_amd.predict import PredictJob, predict_batch, to_host_batch

    # the global batch of batch x world garments (one seed), sharded contiguously: this rank owns garments [lo, hi)
    global_batch = args.batch * world
    hp, sd, shard, (lo, hi) = bench_inputs(args.batch, args.points, args.grid, args.reduce, args.input, rank, world)
    model = ConvImplicitWNFPipeline(**hp)
    model.load_state_dict(sd)
    model = model.to(dev).eval().requires_grad_(False)
    host_data = Batch(sizes=shard.sizes, x=shard.x.pin_memory(), pos=shard.pos.pin_memory(), batch=shard.batch.pin_memory())
    data = host_data.to(dev)                         # resident in HBM before timing
    timer = KernelTimer()
    if args.workload == "pointnet2":
        timer.install_points()
    else:                       # every matrix-core / scan operator of the step is bracketed: the roofline names the one with the most time in it
        timer.install_conv()
        timer.install_decoder()
        timer.install_points()

    # the model's arithmetic is a per-model immutable value (garmentnets_amd/arith.py): each pass installs its own
    headline = Arith.named(args.conv_mode, args.decode_mode, sparse_first_conv=False, winograd=args.winograd == "on", winograd32=args.winograd32 == "on")    # dense, occupancy-independent
    model.arith = headline
    auto_level = [False]

    def step(d=data):
        if args.workload == "pointnet2":
            with torch.no_grad():
                return model.pointnet2_forward(d)
        return predict_batch(model, d, volume_size=args.volume_size, iso_surface_level=0.5, gradient_sigma=0.5, gradient_direction="ascent",
                             auto_level=auto_level[0])

    def step_host_io():
        res = step(host_data.to(dev, non_blocking=True))
        if args.workload == "pointnet2":
            return {k: v.cpu() for k, v in res.items() if torch.is_tensor(v)}
        return to_host_batch(res)

    pipelined = [False]                              # set below: --pipeline-depth 2, full workload, fixed iso level

    def run_steps(fn, n):
        """n steps of fn; with the pipeline on, the same n batches through predict.PredictJob: batch k+1's dense path is queued before
        batch k's host-synchronising tail (vertex counts, mesh slices, surface decode; on a stream of its own) is finished -- every batch
        is begun AND finished inside the call"""
        res = None
        if not (pipelined[0] and fn in (step, step_host_io)):
            for _ in range(n):
                res = fn()
            return res
        prev = None
        for k in range(n):
            d = host_data if fn is step_host_io else data       # a host batch is copied by the job itself, on its front stream
            job = PredictJob(model, d, args.volume_size, 0.5, 0.5, "ascent")
            if prev is not None:
                res = prev.finish(host=fn is step_host_io)
            prev = job
        if prev is not None:
            res = prev.finish(host=fn is step_host_io)
        return res

    sys.path.insert(0, os.path.join(REPO, "tools"))
    from power_trace import HwmonSampler, throttle_read, throttle_window
    hw_passes = {}

    def timed(fn, steps, warmup, hw_name=None):
        thr0 = throttle_read(dev_index) if (hw_name and rank == 0) else None     # (before the warm-up: the GPU goes into the timed steps warm)
        run_steps(fn, warmup)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        timer.reset()
        timer.enabled = True
        sampler = HwmonSampler(dev_index) if (hw_name and rank == 0) else None
        if sampler is not None:
            sampler.__enter__()
        if os.environ.get("GARMENTNETS_BENCH_STALL_TRACE"):      # debugging aid: where is the host when a timed pass stalls (stack of every thread, every 100 ms)
            import faulthandler
            faulthandler.dump_traceback_later(0.1, repeat=True)
        t0 = time.perf_counter()
        res = run_steps(fn, steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if os.environ.get("GARMENTNETS_BENCH_STALL_TRACE"):
            faulthandler.cancel_dump_traceback_later()
            ms_ = torch.cuda.memory_stats()
            print(f"[stall-trace] pass {hw_name}: steps returned after {1e3 * (t1 - t0):.1f} ms, synchronised after {1e3 * dt:.1f} ms; allocator: "
                  f"retries {ms_.get('num_alloc_retries')} device_alloc {ms_.get('num_device_alloc')} device_free {ms_.get('num_device_free')} "
                  f"reserved {ms_.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB (peak {ms_.get('reserved_bytes.all.peak', 0) / 2**30:.1f}) "
                  f"allocated peak {ms_.get('allocated_bytes.all.peak', 0) / 2**30:.1f} GiB", file=sys.stderr, flush=True)
        if sampler is not None:
            sampler.__exit__(None, None, None)
            hw_passes[hw_name] = sampler.summary()
            # why the clock is what it is: share of the pass (warm-up + timed steps: the same workload) the firmware spent limiting it, per limiter
            hw_passes[hw_name]["throttle"] = throttle_window(thr0, throttle_read(dev_index))
        timer.enabled = False
        return dt, res, timer.summary()

    # synthetic weights: use the reference's fixed level 0.5 if every garment's WNF straddles it, else the mid level
    if args.workload == "full":
        probe_res = step()
        if any(bool(torch.isnan(r["verts"]).any()) for r in probe_res):
            auto_level[0] = True
        del probe_res
    if args.pmc_child:                                # under rocprofv3 --pmc (measure_traffic): the headline step, twice, and out
        pmc_child(args, model, data, step)
        return
    pipelined[0] = args.workload == "full" and args.pipeline_depth == 2 and not auto_level[0]
    # warm-up: the W steps asked for, all of them HERE -- the level probe above is an extra untimed step, not one of the W.  (It used to count as
    # one: at W = 1 the timed region then began on an allocator that had seen ONE step, and whether its first timed step found every block
    # cached or paid a 17 GB hipMalloc -- ~0.5 s on these boxes -- came down to which cross-stream frees of the probe had been polled as
    # complete: the same build measured 59 or 143 garments/s depending on it, profiles/r05_ab_experiments.txt section 17.)
    dt, res, groups = timed(step, args.steps, max(2, args.warmup) if pipelined[0] else max(1, args.warmup), hw_name="headline")
    verts_total = None
    checksums = []                                   # per local garment: fp64 sum of its WNF volume (full) / logits (pointnet2), last timed step
    probe = None                                     # slot 0 of the timed result, kept for the oracle check of the cpu_baseline leg
    if args.workload == "full":
        checksums = torch.stack([r["wnf_volume"].double().sum() for r in res]).cpu().tolist()
        verts_total = sum(int(r["verts"].shape[0]) for r in res)
        assert not any(bool(torch.isnan(r["verts"]).any()) for r in res), "marching cubes produced a placeholder mesh"
        if rank == 0:
            r0 = res[0]
            w0 = r0["wnf_volume"]
            level0 = 0.5 * (float(w0.min()) + float(w0.max())) if auto_level[0] else 0.5
            bins0 = torch.round(r0["pred_nocs"] * (model.pointnet2_nocs.nocs_bins - 1)).to(torch.int64).cpu()
            probe = dict(wnf0=w0.cpu(), level0=level0, bins0=bins0, verts0=int(r0["verts"].shape[0]), faces0=int(r0["faces"].shape[0]))
    else:
        checksums = res["per_point_logits"].double().view(hi - lo, -1).sum(dim=1).cpu().tolist()
    if args.workload != "full" and rank == 0:
        n = args.points
        bins0, _, _ = ops.nocs_head(res["per_point_logits"][:n].contiguous(), model.pointnet2_nocs.nocs_bins)
        probe = dict(bins0=bins0.cpu(), feat0=res["per_point_features"][:n].cpu())
    del res

    strict = None
    if not args.no_strict_pass and (args.conv_mode, args.decode_mode) != ("fp32", "fp32") and args.workload == "full":
        model.arith = headline.strict_fp32()
        dt_s, res_s, groups_s = timed(step, args.steps, 1, hw_name="strict_fp32")
        del res_s
        strict = (dt_s, groups_s)
        model.arith = headline
    hostio = None
    if not args.no_host_io_pass:
        dt_h, res_h, _ = timed(step_host_io, args.steps, 1)
        del res_h
        hostio = dt_h

    occupancy = None
    if not args.no_occupancy_pass and args.workload == "full" and args.conv_mode in ("f16x2", "bf16x2"):
        model.arith = headline.replace(sparse_first_conv=True)
        dt_a, _, _ = timed(step, args.steps, 1)
        model.arith = headline
        with torch.no_grad():
            vin = model.volume_agg(model.pointnet2_forward(data)["nocs_data"])
            fl = ops.grid_tile_flags(vin._gn_flat, hi - lo, (args.grid,) * 3)
            occ = int((vin.permute(0, 2, 3, 4, 1) != 0).any(dim=-1).sum())
            if probe is not None:
                probe["occ0"] = (vin[0] != 0).any(dim=0).cpu()
            del vin
        occupancy = {"seconds": dt_a, "occupied_cells_per_garment": occ / (hi - lo), "active_tile_fraction": float(fl.float().mean())}
    elif probe is not None and args.workload == "full":
        with torch.no_grad():
            vin = model.volume_agg(model.pointnet2_forward(data)["nocs_data"])
            probe["occ0"] = (vin[0] != 0).any(dim=0).cpu()
            del vin
    groups_l = None
    literal = None          # the same steps with the GroupNorm shift inside the MFMA operand (Arith.affine_in_weights off): the A/B of DESIGN.md 4.3 on THIS box
    if not args.no_occupancy_pass and args.workload == "full" and args.conv_mode == "f16x2" and headline.affine_in_weights:
        model.arith = headline.replace(affine_in_weights=False)
        literal, _, groups_l = timed(step, args.steps, 1, hw_name="literal_affine")
        model.arith = headline

    # per-stage HIP-event times of ONE extra, untimed step; the stages are the reference's own stage methods.  The same
    # step carries the HIP-event brackets of the HBM-bound members (hbm_members)
    stages_ms = hbm_members = None
    if rank == 0 and args.workload == "full":
        from garmentnets_amd.common import marching_cubes_util as mcu
        hbm = HbmMembers()
        hbm.install()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            for rep in range(2):                      # the second repetition is the one reported (allocator and caches warm)
                hbm.reset()
                ev[0].record()
                p2 = model.pointnet2_forward(data, prefetch_volume=True)
                ev[1].record()
                u3 = model.unet3d_forward(p2)
                ev[2].record()
                wnf_all = model.volume_lattice_forward(u3, args.volume_size)["pred_volume"]
                ev[3].record()
                lvl = 0.5
                if auto_level[0]:
                    mm = torch.stack([wnf_all.min(), wnf_all.max()]).cpu()
                    lvl = 0.5 * (float(mm[0]) + float(mm[1]))
                from garmentnets_amd.predict import _iso_capacity
                job = mcu.IsoBatchJob(args.volume_size, lvl, 0.5, "ascent", cap_v=_iso_capacity(model, args.volume_size))   # what predict_batch does after the lattice
                job.enqueue(wnf_all)
                meshes = job.finish()
                q_all = job.padded_queries()
                if q_all is not None:
                    model.surface_decoder_forward(u3, q_all)
                else:
                    for b_, mesh in enumerate(meshes):
                        if isinstance(mesh, dict):
                            model.surface_decoder_forward(u3.select(b_, b_ + 1), mesh["verts_f32"].view(1, -1, 3))
                ev[4].record()
                torch.cuda.synchronize()
        hbm.uninstall()
        names = ("pointnet2_forward", "unet3d_forward (gridding + UNet)", "volume_lattice_forward (sampler + decoder)", "GGM + MC33 + surface decode")
        stages_ms = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
        hbm_members = hbm.summary(wnf_all, lvl)
        del p2, u3, wnf_all, job, meshes

    fp32_twin = None
    if rank == 0 and args.workload == "full" and args.decode_mode == "f16x2":
        fp32_twin = fp32_twin_accounting(step)

    validation = None
    if rank == 0 and not args.no_validate:
        validation = validate(model, args, dev, auto_level[0])

    in_flight = in_flight_io = None
    if args.workload == "full" and not pipelined[0] and not auto_level[0] and not args.no_in_flight_pass:
        # the same K batches with two in flight (predict.PredictJob).  Last of the GPU passes, behind its own warm-up: a second batch's
        # buffers (tens of GB; hipMalloc of such blocks takes tens of ms each) have to exist in the caching allocator first, and they
        # are released again afterwards
        torch.cuda.empty_cache()
        pipelined[0] = True
        dt_q, res_q, _ = timed(step, args.steps, 5)
        del res_q
        in_flight = dt_q
        # ... and the metric as SURVEY.md 8d words it (H2D of the clouds + D2H of every mesh INSIDE the timed region) with the copies where
        # PredictJob puts them: batch k's meshes go to pinned host memory on the tail stream while batch k+1's dense path runs
        if not args.no_host_io_pass:
            dt_qh, res_qh, _ = timed(step_host_io, args.steps, 2)
            del res_qh
            in_flight_io = dt_qh
        pipelined[0] = False
        torch.cuda.empty_cache()

    lat_b1 = None
    if rank == 0 and world == 1 and args.workload == "full" and not args.no_latency_b1:
        lat_b1 = latency_b1(args, dev)

    # the only collective: per-rank (garments, seconds of each timed pass) over RCCL/xGMI
    n_local = (hi - lo) * args.steps
    per_rank = parallel.gather_metrics([n_local, dt, strict[0] if strict else 0.0, hostio or 0.0, occupancy["seconds"] if occupancy else 0.0,
                                        in_flight or 0.0, literal or 0.0, in_flight_io or 0.0], device=metrics_dev)
    all_sums = parallel.gather_vector(checksums, device=metrics_dev)
    if rank == 0:
        value, tmax = parallel.aggregate_throughput(per_rank)
        garments = sum(r[0] for r in per_rank)
        assert garments == global_batch * args.steps
        split = args.conv_mode != "fp32"
        dtype = "f32" if not split and args.decode_mode == "fp32" else (
            f"f32 ({args.conv_mode} operand split on the 16-bit matrix cores for the 3x3x3 convs" + (" and the decoder MLPs" if args.decode_mode == "f16x2" else "") +
            ", fp32 accumulation; everything else fp32/fp64)" if split else "f32 (f16x2 operand split for the decoder MLPs only)")
        if args.input != "collapsed":
            input_note = ("position-coloured synthetic dress clouds + seeded synthetic checkpoint with the planted NOCS path (synthetic.plant_nocs_path): "
                          "predicted NOCS spread over the garment's shape, realistic occupancy" + (
                              "; planted WNF carrier (synthetic.plant_wnf_path): the 0.5 level set is a thin shell around the garment, a real garment's "
                              "mesh size" if args.input == "planted" else "; WNF = random-weight noise (~10x a real garment's vertices: the tail's stress case)"))
        else:
            input_note = "uniform-colour synthetic dress clouds + un-planted seeded random weights: every cloud collapses into a handful of cells (conditioning study)"
        if args.workload == "pointnet2":
            metric = "garments/s PointNet++ NOCS forward (pointnet2_nocs.py:134-166 + NOCS post-processing)"
            workload = f"PointNet2NOCS forward only, batch={args.batch}/GPU, {args.points}-pt clouds (BASELINE config[1]); input: {input_note}"
            dtype = "f32"
            roofline = points_roofline(groups)
        else:
            metric = "garments/s end-to-end predict (PointNet++ -> gridding -> UNet3D -> WNF decode -> marching cubes)"
            workload = (f"full conv_implicit_wnf pipeline, batch={args.batch}/GPU, {args.points}-pt clouds, {args.grid}^3 feature volume ({args.reduce}), "
                        f"{args.volume_size}^3 WNF + GGM + MC33 + surface decode; input: {input_note}"
                        + (f" ({occupancy['occupied_cells_per_garment']:.0f} occupied cells per garment)" if occupancy else ""))
            if args.no_pmc or world > 1:          # (the counter passes are a single-GPU measurement: at N > 1 the line carries null)
                traffic = (None, "--no-pmc" if args.no_pmc else "not collected at N > 1 (single-GPU measurement: run bench.py --gpus 1)")
            else:
                traffic = measure_traffic(args)
            roofline = pick_roofline(args, groups, args.conv_mode, args.decode_mode, hw_passes.get("headline"), traffic)
        line = {
            "metric": metric, "value": garments / tmax, "unit": "garments/s", "input": args.input,
            "mesh_verts_per_garment": (verts_total / (hi - lo)) if verts_total is not None else None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "dtype_short": "f32" if dtype == "f32" else f"f32 ({args.conv_mode if split else 'fp32'} convs, {args.decode_mode} decoder MLPs, fp32 accumulation)",
            "data": "synthetic",
            "config": {"workload": workload, "input": args.input, "batch_per_gpu": args.batch, "global_batch": global_batch,
                       "sharding": f"garments [r*{args.batch}, (r+1)*{args.batch}) of one seeded global batch per rank (parallel.shard_range)",
                       "points": args.points, "grid": args.grid, "reduce": args.reduce,
                       "volume_size": args.volume_size, "iso_level": "mid(min,max)" if auto_level[0] else 0.5,
                       "weights": "seeded synthetic (reference architecture)" + (" + planted NOCS path" if args.input != "collapsed" else "") + (
                           " + planted WNF carrier path" if args.input == "planted" else ""),
                       "wnf_input": {"planted": "garment-like: smooth shell around the garment (synthetic.plant_wnf_path)", "noisy_wnf": "random-weight noise (stress case)",
                                     "collapsed": "random-weight noise"}[args.input],
                       "mesh_verts_per_garment": (verts_total / (hi - lo)) if verts_total is not None else None,
                       "encoder_convs": "dense: every tile through the matrix cores (occupancy-aware launch OFF for the headline)" + (
                           "; GroupNorm affine of the two convolutions behind the scattered volume folded into per-sample weights + a bias table, operand exactly zero in "
                           "empty cells (Arith.affine_in_weights; the literal form is timed as literal_affine)" if headline.affine_in_weights and args.conv_mode == "f16x2" else "") + (
                           "; 128-wide convolutions in Winograd F(2,3) form along x (Arith.winograd: 36 of 54 tap products; --winograd off = the direct form)"
                           if headline.winograd and args.conv_mode == "f16x2" else ""), "mesh_verts_per_step": verts_total,
                       "parallelism": f"dp{world} (independent garment shards, no data-path collective)"},
            "timed_region": "inputs resident in HBM, results left on the device (with_host_io adds H2D of the clouds + D2H of every mesh); K batches "
                            "begun and finished between the two barriers" + (
                                ", two in flight: batch k+1's PointNet++/UNet/lattice is queued before batch k's tail (vertex counts to the host, mesh "
                                "slices, surface decode) is finished (predict.PredictJob; bit-equal results: tests/test_gpu_api.py)" if pipelined[0] else ", one at a time"),
            "pipeline_depth": 2 if pipelined[0] else 1,
            "rccl_ranks_seen": len(per_rank),
            "dist_backend": backend if world > 1 else None,
            "per_rank": {"seconds": [r[1] for r in per_rank], "garments_per_s": [r[0] / r[1] for r in per_rank],
                         "slowest_over_fastest": max(r[1] for r in per_rank) / min(r[1] for r in per_rank),
                         "host_affinity_rank0": affinity},
            "scaling_vs_n1": (garments / tmax) / (world * args.n1_value) if args.n1_value else None,
            "garment_checksums": [c for r in all_sums for c in r],      # global garment order (rank-major): fp64 sum of each result
            "stages_ms": stages_ms,
            "roofline": roofline,
        }
        if hbm_members is not None:
            line["hbm_members"] = hbm_members
        if fp32_twin is not None:
            line["fp32_twin"] = fp32_twin
        if lat_b1 is not None:
            line["latency_b1"] = lat_b1
        if in_flight is not None:
            tq = max(r[5] for r in per_rank)
            line["two_in_flight"] = {"value": garments / tq, "unit": "garments/s", "ms_per_step": 1e3 * tq / args.steps, "steps": args.steps,
                                     "what": "the same K batches through predict.PredictJob: batch k+1's dense path is queued before batch k's tail (vertex counts to "
                                             "the host, mesh slices, surface decode; on its own stream) is finished, and batch k+1's PointNet++ (serial farthest-point "
                                             "sampling) runs on a front stream beside batch k's UNet.  Bit-equal results (tests/test_gpu_api.py)"}
        if in_flight_io is not None:
            tqh = max(r[7] for r in per_rank)
            line["two_in_flight_with_host_io"] = {
                "value": garments / tqh, "unit": "garments/s", "ms_per_step": 1e3 * tqh / args.steps, "steps": args.steps,
                "fraction_of_headline": (garments / tqh) / (garments / tmax),
                "what": "SURVEY.md 8d's metric (pinned-host -> HBM copy of the clouds ... device -> pinned-host copy of every garment's verts / faces / "
                        "normals / values / gradient magnitude / warp field inside the timed region) with two batches in flight: batch k's meshes are "
                        "copied on PredictJob's tail stream (finish(host=True)) while batch k+1's PointNet++ / UNet / lattice run on the main stream"}
        if strict:
            ts = max(r[2] for r in per_rank)
            line["strict_fp32"] = {"value": garments / ts, "unit": "garments/s", "ms_per_step": 1e3 * ts / args.steps, "steps": args.steps,
                                   "dtype": "f32 (v_mfma_f32_32x32x2_f32 convs + fp32 decoder MLPs: --conv-mode fp32 --decode-mode fp32)",
                                   "roofline": pick_roofline(args, strict[1], "fp32", "fp32", hw_passes.get("strict_fp32"))}
        if hostio:
            th = max(r[3] for r in per_rank)
            line["with_host_io"] = {"value": garments / th, "unit": "garments/s", "ms_per_step": 1e3 * th / args.steps, "steps": args.steps,
                                    "includes": "pinned-host -> HBM copy of the clouds, the step, device -> host copy of verts / faces / normals / values / "
                                                "gradient magnitude / warp field of every garment (predict.to_host)" if args.workload == "full" else
                                                "pinned-host -> HBM copy of the clouds, the step, device -> host copy of every result tensor"}
        if occupancy:
            to = max(r[4] for r in per_rank)
            line["occupancy_aware"] = {
                "value": garments / to, "unit": "garments/s", "ms_per_step": 1e3 * to / args.steps,
                "what": "the library default: the first two UNet convolutions visit only the output tiles that can see an occupied cell (exact, bit-identical "
                        "to the dense launch; switched off for the headline value) -- same input, same K steps",
                "occupied_cells_per_garment": occupancy["occupied_cells_per_garment"], "active_tile_fraction": occupancy["active_tile_fraction"]}
        if literal:
            tl = max(r[6] for r in per_rank)
            line["literal_affine"] = {
                "value": garments / tl, "unit": "garments/s", "ms_per_step": 1e3 * tl / args.steps,
                "what": "the same K steps with Arith.affine_in_weights off: the GroupNorm shift inside the MFMA operand of the first two encoder convolutions "
                        "(every voxel of the >= 99.7 % empty volume non-zero) instead of in per-sample weights + a bias table -- the same MACs through the same "
                        "kernels; the difference is clock under the socket's power cap (DESIGN.md 4.3).  This is the occupancy-INDEPENDENT "
                        "figure: quote it next to the headline",
                "roofline": pick_roofline(args, groups_l, args.conv_mode, args.decode_mode, hw_passes.get("literal_affine"))}
        if validation is not None:
            line["validation"] = validation
        if world == 1 and not args.no_cpu_baseline:
            if full_mask is not None:
                os.sched_setaffinity(0, full_mask)          # the CPU baseline may use every core of the host
            line["cpu_baseline"], check = cpu_baseline(args, hp, sd, shard, probe)
            if check is not None:
                line["oracle_check"] = check
        emit(line)
        if validation is not None and not validation["ok"]:
            raise SystemExit("bench.py: validation failed (identical garments gave different results in slot 0 and the last slot)")
        if line.get("oracle_check") is not None and not line["oracle_check"]["ok"]:
            raise SystemExit("bench.py: oracle check failed (slot 0 of the timed result differs from the CPU oracle beyond the tolerance)")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
