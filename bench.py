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
                        metric as SURVEY.md 8d words it
  roofline              dominant kernel by time: launches bracketed with HIP events on the launch stream during the timed steps,
                        labelled with the kernel variant the C ABI reports having launched (gn_last_kernel), achieved = algorithmic
                        FLOPs (54*Cin*Cout per voxel) / time; `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
                        of this same command (profiles/, `traffic_source`), only when the workload is the profiled one
  occupancy_aware       the same K steps with the library's default occupancy-aware first UNet convolution (exact: bit-identical outputs,
                        tests/test_gpu_parity.py::test_sparse_first_conv_is_bit_identical_to_dense) on (a) the same synthetic clouds -- whose
                        predicted NOCS coordinates collapse into a handful of cells under seeded random weights, so this is a best case --
                        and (b) clouds whose NOCS coordinates are the garment's own normalised, 64-bin quantised positions (the
                        occupancy a trained PointNet++ produces: thousands of cells), with the dense figure for (b) next to it.  The
                        HEADLINE value is measured with the occupancy-aware path switched OFF: every tile goes through the matrix cores
                        and the number does not depend on where the points fall
  validation            untimed: a batch of IDENTICAL garments (PointConv self-loop quirk off) must give the same WNF and mesh in the
                        first and the last slot -- garbage in the upper slots of the benchmark batch cannot go unnoticed
  cpu_baseline          the CPU oracle (torch-CPU port of the reference path) timed on this host for a bounded sample
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
    ap.add_argument("--conv-mode", default="f16x2", choices=["f16x2", "fp32", "bf16x3", "bf16x2"],
                    help="arithmetic of the 3x3x3 convs of the HEADLINE pass: f16x2 (default; fp32 operands split into two fp16 planes, fp32 "
                         "accumulation), fp32 (v_mfma_f32_32x32x2_f32), bf16x3, bf16x2 (preview quality)")
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
            name, work, byts = describe(out, *a, **kw)
            timer.records.append((name, work, byts, e0, e1))
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

        ops.conv3d_gcr = self._wrap(ops.conv3d_gcr, describe)
        ops.conv3d_gcr_split = self._wrap(ops.conv3d_gcr_split, describe)
        ops.upconv_partial = self._wrap(ops.upconv_partial, describe_up)

    def install_points(self):
        """PointNet++ operators (config[1]): work = squared-distance evaluations for fps / ball query / kNN, FLOPs for the GEMMs"""
        from garmentnets_amd import ops

        def d_fps(res_, pos, ptr, out_ptr, max_points, m_total, *r, **k):
            B = ptr.numel() - 1
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


def measured_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs
    of THIS command, corrected as profiles/*_fetch_calibration.txt documents; tools/pmc_summary.py) -- only quoted when the workload is
    the one that was profiled.  -> (bytes or None, source file or None)"""
    if (args.workload, args.batch, args.points, args.grid, args.reduce, args.volume_size, args.conv_mode) != ("full", 16, 6000, 128, "mean", 128, "f16x2"):
        return None, None
    for name in ("r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            k = json.load(open(path))["kernels"].get(kernel)
            if k is not None:
                return k["hbm_bytes"], "profiles/" + name + " (committed PMC passes of this command, not this run)"
    return None, None


def conv_roofline(args, groups, conv_mode):
    key = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[key]
    achieved = g["work"] / (g["ms"] * 1e-3) / 1e12          # algorithmic (fp32) FLOPs: 54*Cin*Cout per voxel
    if conv_mode == "fp32":
        peak, peak_note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense peak"
    else:
        n = SPLIT_PRODUCTS[conv_mode]
        peak = PEAK_16BIT_MFMA_TFLOPS / n
        peak_note = (f"16-bit MFMA dense peak {PEAK_16BIT_MFMA_TFLOPS:.0f} / {n} matrix-core products per algorithmic fp32 product "
                     f"({conv_mode}); executed {achieved * n:.0f} TFLOP/s; the fp32-MFMA peak is {PEAK_FP32_MFMA_TFLOPS}")
    traffic, src = measured_traffic(args, key) if conv_mode == args.conv_mode else (None, None)
    return {"bound": "mfma", "kernel": key, "kernel_label": "reported by the C ABI (gn_last_kernel) after each launch",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
            "traffic": traffic, "traffic_source": src,
            "launches": g["n"], "avg_launch_ms": g["ms"] / g["n"], "flops_per_launch": g["work"] / g["n"],
            "algorithmic_bytes_per_launch": g["bytes"] / g["n"],
            "hbm_frac_of_8TBs": g["bytes"] / (g["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "all_conv_instances": {k: {"launches": v["n"], "ms": v["ms"], "tflops": v["work"] / (v["ms"] * 1e-3) / 1e12}
                                   for k, v in groups.items()}}


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


def cpu_baseline(args, hp, sd):
    """The oracle (a torch-CPU port of the reference path: 'port') on this host's cores, bounded sample."""
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
    if args.workload == "pointnet2":
        n = max(n, 4)
        x, pos, batch = S.synthetic_cloud(n, args.points, seed=12345)
        with torch.no_grad():
            P.pointnet2_forward(sd_cpu, hp, x[:args.points], pos[:args.points], batch[:args.points])
            t0 = time.time()
            P.pointnet2_forward(sd_cpu, hp, x, pos, batch)
            dt = time.time() - t0
        what = f"{n} garments, PointNet2NOCS forward + NOCS post-processing (N={args.points})"
    else:
        x, pos, batch = S.synthetic_cloud(n, args.points, seed=12345)
        t0 = time.time()
        P.predict(sd_cpu, hp, x, pos, batch, Q=args.volume_size, level=0.5, sigma=0.5, auto_level=True)
        dt = time.time() - t0
        what = f"{n} garment(s) of the same workload (N={args.points}, G={args.grid} {args.reduce}, Q={args.volume_size})"
    return {"value": n / dt, "unit": "garments/s", "cores": cores, "kind": "port",
            "sample": f"{what}, oracle/pipeline.py on torch-CPU fp32 with {cores} of {ncpu} hardware threads (fastest of a short sweep; fps / ball "
                      f"query / kNN / GGM / marching cubes single-threaded C as in the reference), {dt:.1f} s"}


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
        x, pos, _ = S.synthetic_cloud(1, n, seed=4242)
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
    # (GARMENTNETS_DIST_BACKEND=gloo: control-flow smoke test of the N > 1 path on a box with fewer GPUs than ranks -- ranks share devices)
    backend = os.environ.get("GARMENTNETS_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    parallel.init(backend=backend, device=dev)     # "nccl" is RCCL on ROCm; no-op for one process
    metrics_dev = dev if backend == "nccl" else "cpu"

    from garmentnets_amd import ops, synthetic as S
    from garmentnets_amd.batch import Batch
    from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
    from garmentnets_amd.predict import PredictJob, predict_batch, to_host

    hp = S.default_hparams(grid=args.grid, reduce_method=args.reduce)
    sd = S.synthetic_state_dict(hp, 0)
    model = ConvImplicitWNFPipeline(**hp)
    model.load_state_dict(sd)
    model = model.to(dev).eval().requires_grad_(False)
    # the global batch of batch x world garments (one seed), sharded contiguously: this rank owns garments [lo, hi)
    global_batch = args.batch * world
    shard, (lo, hi) = parallel.shard_batch(global_batch, args.points, CLOUD_SEED, rank, world)
    host_data = Batch(sizes=shard.sizes, x=shard.x.pin_memory(), pos=shard.pos.pin_memory(), batch=shard.batch.pin_memory())
    data = host_data.to(dev)                         # resident in HBM before timing
    timer = KernelTimer()
    if args.workload == "pointnet2":
        timer.install_points()
    else:
        timer.install_conv()
    import garmentnets_amd.components.unet3d as u
    u.ops = ops

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
        return [to_host(r) for r in res]

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
            d = host_data.to(dev, non_blocking=True) if fn is step_host_io else data
            job = PredictJob(model, d, args.volume_size, 0.5, 0.5, "ascent", bank=1 + (k & 1))
            if prev is not None:
                res = prev.finish(host=fn is step_host_io)
            prev = job
        if prev is not None:
            res = prev.finish(host=fn is step_host_io)
        return res

    def timed(fn, steps, warmup):
        run_steps(fn, warmup)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        timer.reset()
        timer.enabled = True
        t0 = time.perf_counter()
        res = run_steps(fn, steps)
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
        return dt, res, timer.summary()

    def set_modes(conv, decode):
        ops.CONV_MODE = ops.CONV_MODE_NAMES[conv]
        ops.DECODE_MODE = decode

    set_modes(args.conv_mode, args.decode_mode)
    ops.SPARSE_FIRST_CONV = False                    # headline / strict / host-io passes: dense, occupancy-independent
    # synthetic weights: use the reference's fixed level 0.5 if every garment's WNF straddles it, else the mid level
    if args.workload == "full":
        probe = step()
        if any(bool(torch.isnan(r["verts"]).any()) for r in probe):
            auto_level[0] = True
        del probe
    pipelined[0] = args.workload == "full" and args.pipeline_depth == 2 and not auto_level[0]
    dt, res, groups = timed(step, args.steps, max(2, args.warmup - 1) if pipelined[0] else max(0, args.warmup - (1 if args.workload == "full" else 0)))
    verts_total = None
    if args.workload == "full":
        verts_total = sum(int(r["verts"].shape[0]) for r in res)
        assert not any(bool(torch.isnan(r["verts"]).any()) for r in res), "marching cubes produced a placeholder mesh"
    del res

    strict = None
    if not args.no_strict_pass and (args.conv_mode, args.decode_mode) != ("fp32", "fp32") and args.workload == "full":
        set_modes("fp32", "fp32")
        dt_s, res_s, groups_s = timed(step, args.steps, 1)
        del res_s
        strict = (dt_s, groups_s)
        set_modes(args.conv_mode, args.decode_mode)
    hostio = None
    if not args.no_host_io_pass:
        dt_h, res_h, _ = timed(step_host_io, args.steps, 1)
        del res_h
        hostio = dt_h

    occupancy = None
    if not args.no_occupancy_pass and args.workload == "full" and args.conv_mode in ("f16x2", "bf16x2"):
        def tiles_seen():
            with torch.no_grad():
                vin = model.volume_agg(model.pointnet2_forward(data)["nocs_data"])
                fl = ops.grid_tile_flags(vin._gn_flat, hi - lo, (args.grid,) * 3)
                occ = int((vin.permute(0, 2, 3, 4, 1) != 0).any(dim=-1).sum())
            return {"occupied_cells_per_garment": occ / (hi - lo), "active_tile_fraction": float(fl.float().mean())}

        occupancy = {}
        ops.SPARSE_FIRST_CONV = True
        dt_a, _, _ = timed(step, args.steps, 1)
        occupancy["synthetic_clouds"] = dict(tiles_seen(), seconds=dt_a)
        orig_p2 = model.pointnet2_forward

        def spread_nocs(d):                          # NOCS := the garment's own normalised, 64-bin quantised positions (bench input, not the network's prediction)
            res = orig_p2(d)
            p3 = d.pos.view(hi - lo, args.points, 3)
            mn, mx = p3.min(dim=1, keepdim=True)[0], p3.max(dim=1, keepdim=True)[0]
            res["nocs_data"].pos = (torch.round((0.1 + 0.8 * (p3 - mn) / (mx - mn)) * 63) * (1.0 / 63)).view(-1, 3).contiguous()
            return res
        model.pointnet2_forward = spread_nocs
        try:
            dt_b, _, _ = timed(step, args.steps, 1)
            occupancy["realistic_occupancy"] = dict(tiles_seen(), seconds=dt_b)
            ops.SPARSE_FIRST_CONV = False
            dt_c, _, _ = timed(step, args.steps, 1)
            occupancy["realistic_occupancy"]["seconds_dense"] = dt_c
        finally:
            del model.pointnet2_forward
            ops.SPARSE_FIRST_CONV = False

    # per-stage HIP-event times of ONE extra, untimed step; the stages are the reference's own stage methods
    stages_ms = None
    if rank == 0 and args.workload == "full":
        from garmentnets_amd.common import marching_cubes_util as mcu
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            ev[0].record()
            p2 = model.pointnet2_forward(data)
            ev[1].record()
            u3 = model.unet3d_forward(p2)
            ev[2].record()
            wnf_all = model.volume_lattice_forward(u3, args.volume_size)["pred_volume"]
            ev[3].record()
            lvl = 0.5
            if auto_level[0]:
                mm = torch.stack([wnf_all.min(), wnf_all.max()]).cpu()
                lvl = 0.5 * (float(mm[0]) + float(mm[1]))
            job = mcu.IsoBatchJob(args.volume_size, lvl, 0.5, "ascent")          # what predict_batch does after the lattice
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
        names = ("pointnet2_forward", "unet3d_forward (gridding + UNet)", "volume_lattice_forward (sampler + decoder)", "GGM + MC33 + surface decode")
        stages_ms = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
        del p2, u3, wnf_all

    validation = None
    if rank == 0 and not args.no_validate:
        validation = validate(model, args, dev, auto_level[0])

    in_flight = None
    if args.workload == "full" and not pipelined[0] and not auto_level[0] and not args.no_in_flight_pass:
        # the same K batches with two in flight (predict.PredictJob).  Last of the GPU passes, behind its own warm-up: a second batch's
        # buffers (tens of GB; hipMalloc of such blocks takes tens of ms each) have to exist in the caching allocator first, and they
        # are released again afterwards
        torch.cuda.empty_cache()
        pipelined[0] = True
        dt_q, res_q, _ = timed(step, args.steps, 5)
        del res_q
        in_flight = dt_q
        pipelined[0] = False
        torch.cuda.empty_cache()

    # the only collective: per-rank (garments, seconds of each timed pass) over RCCL/xGMI
    n_local = (hi - lo) * args.steps
    occ_t = [occupancy[k].get(f, 0.0) for k, f in (("synthetic_clouds", "seconds"), ("realistic_occupancy", "seconds"), ("realistic_occupancy", "seconds_dense"))] \
        if occupancy else [0.0, 0.0, 0.0]
    per_rank = parallel.gather_metrics([n_local, dt, strict[0] if strict else 0.0, hostio or 0.0] + occ_t + [in_flight or 0.0], device=metrics_dev)
    if rank == 0:
        value, tmax = parallel.aggregate_throughput(per_rank)
        garments = sum(r[0] for r in per_rank)
        assert garments == global_batch * args.steps
        split = args.conv_mode != "fp32"
        dtype = "f32" if not split and args.decode_mode == "fp32" else (
            f"f32 ({args.conv_mode} operand split on the 16-bit matrix cores for the 3x3x3 convs" + (" and the decoder MLPs" if args.decode_mode == "f16x2" else "") +
            ", fp32 accumulation; everything else fp32/fp64)" if split else "f32 (f16x2 operand split for the decoder MLPs only)")
        if args.workload == "pointnet2":
            metric = "garments/s PointNet++ NOCS forward (pointnet2_nocs.py:134-166 + NOCS post-processing)"
            workload = f"PointNet2NOCS forward only, batch={args.batch}/GPU, {args.points}-pt clouds (BASELINE config[1])"
            dtype = "f32"
            roofline = points_roofline(groups)
        else:
            metric = "garments/s end-to-end predict (PointNet++ -> gridding -> UNet3D -> WNF decode -> marching cubes)"
            workload = (f"full conv_implicit_wnf pipeline, batch={args.batch}/GPU, {args.points}-pt clouds, {args.grid}^3 feature volume ({args.reduce}), "
                        f"{args.volume_size}^3 WNF + GGM + MC33 + surface decode")
            roofline = conv_roofline(args, groups, args.conv_mode)
        line = {
            "metric": metric, "value": garments / tmax, "unit": "garments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": args.batch, "global_batch": global_batch,
                       "sharding": f"garments [r*{args.batch}, (r+1)*{args.batch}) of one seeded global batch per rank (parallel.shard_range)",
                       "points": args.points, "grid": args.grid, "reduce": args.reduce,
                       "volume_size": args.volume_size, "iso_level": "mid(min,max)" if auto_level[0] else 0.5,
                       "weights": "seeded synthetic (reference architecture)", "mesh_verts_per_step": verts_total,
                       "parallelism": f"dp{world} (independent garment shards, no data-path collective)"},
            "timed_region": "inputs resident in HBM, results left on the device (with_host_io adds H2D of the clouds + D2H of every mesh); K batches "
                            "begun and finished between the two barriers" + (
                                ", two in flight: batch k+1's PointNet++/UNet/lattice is queued before batch k's tail (vertex counts to the host, mesh "
                                "slices, surface decode) is finished (predict.PredictJob; bit-equal results: tests/test_gpu_api.py)" if pipelined[0] else ", one at a time"),
            "pipeline_depth": 2 if pipelined[0] else 1,
            "rccl_ranks_seen": len(per_rank),
            "stages_ms": stages_ms,
            "roofline": roofline,
        }
        if in_flight is not None:
            tq = max(r[7] for r in per_rank)
            line["two_in_flight"] = {"value": garments / tq, "unit": "garments/s", "ms_per_step": 1e3 * tq / args.steps, "steps": args.steps,
                                     "what": "the same K batches through predict.PredictJob: batch k+1's PointNet++ / UNet / lattice is queued before batch "
                                             "k's tail (vertex counts to the host, mesh slices, surface decode; on its own stream) is finished.  Bit-equal "
                                             "results (tests/test_gpu_api.py); the step is matrix-core / power bound, so hiding the latency-bound tail buys "
                                             "little"}
        if strict:
            ts = max(r[2] for r in per_rank)
            line["strict_fp32"] = {"value": garments / ts, "unit": "garments/s", "ms_per_step": 1e3 * ts / args.steps, "steps": args.steps,
                                   "dtype": "f32 (v_mfma_f32_32x32x2_f32 convs + fp32 decoder MLPs: --conv-mode fp32 --decode-mode fp32)",
                                   "roofline": conv_roofline(args, strict[1], "fp32")}
        if hostio:
            th = max(r[3] for r in per_rank)
            line["with_host_io"] = {"value": garments / th, "unit": "garments/s", "ms_per_step": 1e3 * th / args.steps, "steps": args.steps,
                                    "includes": "pinned-host -> HBM copy of the clouds, the step, device -> host copy of verts / faces / normals / values / "
                                                "gradient magnitude / warp field of every garment (predict.to_host)" if args.workload == "full" else
                                                "pinned-host -> HBM copy of the clouds, the step, device -> host copy of every result tensor"}
        if occupancy:
            def rate(slot):
                t = max(r[slot] for r in per_rank)
                return {"value": garments / t, "unit": "garments/s", "ms_per_step": 1e3 * t / args.steps}
            line["occupancy_aware"] = {
                "what": "first UNet convolution visits only the output tiles that can see an occupied cell (exact, bit-identical to the dense launch; the "
                        "library default, switched off for the headline value)",
                "synthetic_clouds": dict(rate(4), **{k: v for k, v in occupancy["synthetic_clouds"].items() if k != "seconds"},
                                         note="seeded random weights collapse every cloud's NOCS prediction into a handful of cells: best case, not representative"),
                "realistic_occupancy": dict(rate(5), dense=rate(6), **{k: v for k, v in occupancy["realistic_occupancy"].items() if not k.startswith("seconds")},
                                            note="NOCS coordinates := the garment's own normalised, 64-bin quantised point positions (bench input construction), "
                                                 "i.e. the cell occupancy a trained PointNet++ produces")}
        if validation is not None:
            line["validation"] = validation
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, hp, sd)
        print(json.dumps(line))
        if validation is not None and not validation["ok"]:
            raise SystemExit("bench.py: validation failed (identical garments gave different results in slot 0 and the last slot)")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
