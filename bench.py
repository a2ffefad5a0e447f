#!/usr/bin/env python3
"""bench.py -- end-to-end GarmentNets inference throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the whole hot path (predict.py:138-209: PointNet++ -> gridding -> 3-D UNet -> (Q,Q,Q) WNF
decode -> Gaussian gradient magnitude -> Lewiner marching cubes -> surface decode) over one batch of synthetic garments
that is already resident in HBM.  Every rank owns its own batch (weak scaling, garments never move between GPUs);
the only collective is an all-gather of per-rank timings (RCCL).  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the 3x3x3 conv (conv3d_gcr_kernel, fp32 MFMA).  Its launches are bracketed with HIP
events on the launch stream during the timed steps; achieved = algorithmic FLOPs (54*Cin*Cout*voxels per launch) / time.
cpu_baseline: the CPU oracle (torch-CPU port of the reference path) timed on this host for a bounded sample.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="garments per GPU per step")
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--grid", type=int, default=128, help="feature-volume edge G (north_star: 128; reference ckpt default: 32)")
    ap.add_argument("--reduce", default="mean", choices=["mean", "max"])
    ap.add_argument("--volume-size", type=int, default=128, help="WNF query volume edge Q")
    ap.add_argument("--conv-mode", default="f16x2", choices=["f16x2", "fp32", "bf16x3", "bf16x2"],
                    help="arithmetic of the 3x3x3 convs: f16x2 (default; fp32 operands split into two fp16 planes, fp32 accumulation, "
                         "error vs fp64 below the fp32 kernel's), fp32 (v_mfma_f32_32x32x2_f32), bf16x3, bf16x2 (preview quality)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-garments", type=int, default=1)
    return ap.parse_args()


class ConvTimer:
    """HIP-event brackets around every conv3d launch (on torch's current stream = the launch stream)."""

    def __init__(self):
        self.records = []   # (kernel, flops, bytes, start_event, end_event)
        self.enabled = False

    def install(self):
        from garmentnets_amd import ops
        timer = self

        def wrap(orig, kernel_name):
            def timed(src0, src1, a, d, wp, cout, relu=True, with_stats=False):
                if not timer.enabled:
                    return orig(src0, src1, a, d, wp, cout, relu, with_stats)
                B, D, H, W, C0 = src0.shape
                cin = C0 + (0 if src1 is None else src1.shape[-1])
                tiles = -(-D // 4) * -(-H // 8) * -(-W // 8)
                nt = 2 if (cout % 64 == 0 and tiles * (cout // 64) * B >= 1024) else 1
                if hasattr(wp, "mode") and wp.mode != 3 and cout % 128 == 0 and cin <= 384 and tiles * (cout // 128) * B >= 512:
                    nt = 4                                  # the 128-wide variant (same dispatch rule as gn_conv3d_gcr_split)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig(src0, src1, a, d, wp, cout, relu, with_stats)
                e1.record()
                wbytes = (wp.tensor.numel() * 2.0) if hasattr(wp, "tensor") else wp.numel() * 4.0
                timer.records.append((kernel_name(nt, wp), 54.0 * cin * cout * B * D * H * W, (cin + cout) * 4.0 * B * D * H * W + wbytes, e0, e1))
                return out
            return timed

        ops.conv3d_gcr = wrap(ops.conv3d_gcr, lambda nt, wp: f"conv3d_gcr_kernel<{nt}>")
        ops.conv3d_gcr_split = wrap(ops.conv3d_gcr_split, lambda nt, wp: ("conv3d_split_wide_kernel<2, %s>" if nt == 4 else "conv3d_split_kernel<%d, %d, %%s, 1>" % (
            nt, 3 if wp.mode == ops.SPLIT_BF16X3 else 2)) % ("true" if wp.mode == ops.SPLIT_F16X2 else "false"))
        import garmentnets_amd.components.unet3d as u
        u.ops = ops

    def summary(self):
        groups = {}
        for name, flops, byts, e0, e1 in self.records:
            g = groups.setdefault(name, dict(flops=0.0, bytes=0.0, ms=0.0, n=0))
            g["flops"] += flops
            g["bytes"] += byts
            g["ms"] += e0.elapsed_time(e1)
            g["n"] += 1
        return groups


def measured_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs of THIS command, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_summary.py) -- only quoted when
    the workload is the one that was profiled."""
    path = os.path.join(REPO, "profiles", "r01_hbm_traffic.json")
    if not (os.path.exists(path) and (args.batch, args.points, args.grid, args.reduce, args.volume_size, args.conv_mode) == (16, 6000, 128, "mean", 128, "f16x2")):
        return None
    k = json.load(open(path))["kernels"].get(kernel)
    return None if k is None else k["hbm_bytes"]


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
    x, pos, batch = S.synthetic_cloud(n, args.points, seed=12345)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    t0 = time.time()
    P.predict(sd_cpu, hp, x, pos, batch, Q=args.volume_size, level=0.5, sigma=0.5, auto_level=True)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "garments/s", "cores": cores, "kind": "port",
            "sample": f"{n} garment(s) of the same workload (N={args.points}, G={args.grid} {args.reduce}, Q={args.volume_size}), "
                      f"oracle/pipeline.py on torch-CPU fp32 with {cores} of {ncpu} hardware threads (fastest of a short sweep; GGM / marching cubes "
                      f"single-threaded C as in the reference), {dt:.1f} s"}


def main():
    args = parse()
    from garmentnets_amd import parallel
    rank, local_rank, world = parallel.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    parallel.init(backend="nccl", device=dev)      # "nccl" is RCCL on ROCm; no-op for one process

    from garmentnets_amd import synthetic as S
    from garmentnets_amd.batch import Batch
    from garmentnets_amd.networks.conv_implicit_wnf import ConvImplicitWNFPipeline
    from garmentnets_amd.predict import predict_batch

    hp = S.default_hparams(grid=args.grid, reduce_method=args.reduce)
    sd = S.synthetic_state_dict(hp, 0)
    model = ConvImplicitWNFPipeline(**hp)
    model.load_state_dict(sd)
    model = model.to(dev).eval().requires_grad_(False)
    x, pos, batch = S.synthetic_cloud(args.batch, args.points, seed=1000 * rank)
    data = Batch(sizes=[args.points] * args.batch, x=x, pos=pos, batch=batch).to(dev)   # resident in HBM before timing
    timer = ConvTimer()
    timer.install()
    from garmentnets_amd import ops as _ops
    _ops.CONV_MODE = _ops.CONV_MODE_NAMES[args.conv_mode]

    def step():
        return predict_batch(model, data, volume_size=args.volume_size, iso_surface_level=0.5, gradient_sigma=0.5,
                             gradient_direction="ascent", auto_level=auto_level)

    # synthetic weights: use the reference's fixed level 0.5 if every garment's WNF straddles it, else the mid level
    auto_level = False
    probe = step()
    if any(bool(torch.isnan(r["verts"]).any()) for r in probe):
        auto_level = True
    verts_total = 0
    for _ in range(max(0, args.warmup - 1)):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    verts_total = sum(int(r["verts"].shape[0]) for r in res)
    assert not any(bool(torch.isnan(r["verts"]).any()) for r in res), "marching cubes produced a placeholder mesh"

    # per-stage HIP-event times of ONE extra, untimed step; the stages are the reference's own stage methods
    stages_ms = None
    if rank == 0:
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
            if auto_level:
                mm = torch.stack([wnf_all.min(), wnf_all.max()]).cpu()
                lvl = 0.5 * (float(mm[0]) + float(mm[1]))
            for b_, mesh in enumerate(mcu.wnf_batch_to_meshes_gpu(wnf_all, lvl, 0.5, "ascent")):
                if isinstance(mesh, dict):
                    model.surface_decoder_forward(u3.select(b_, b_ + 1), mesh["verts_f32"].view(1, -1, 3))
            ev[4].record()
        torch.cuda.synchronize()
        names = ("pointnet2_forward", "unet3d_forward (gridding + UNet)", "volume_lattice_forward (sampler + decoder)", "GGM + MC33 + surface decode")
        stages_ms = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}

    # the only collective: per-rank (garments, seconds) over RCCL/xGMI
    per_rank = parallel.gather_metrics([args.batch * args.steps, dt], device=dev)
    if rank == 0:
        value, tmax = parallel.aggregate_throughput(per_rank)
        garments = args.batch * world * args.steps
        groups = timer.summary()
        key = max(groups, key=lambda k: groups[k]["ms"])
        g = groups[key]
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12          # algorithmic (fp32) FLOPs: 54*Cin*Cout per voxel
        if args.conv_mode == "fp32":
            peak, peak_note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense peak"
        else:
            n = SPLIT_PRODUCTS[args.conv_mode]
            peak = PEAK_16BIT_MFMA_TFLOPS / n
            peak_note = (f"16-bit MFMA dense peak {PEAK_16BIT_MFMA_TFLOPS:.0f} / {n} matrix-core products per algorithmic fp32 product "
                         f"({args.conv_mode}); executed {achieved * n:.0f} TFLOP/s; the fp32-MFMA peak is {PEAK_FP32_MFMA_TFLOPS}")
        line = {
            "metric": "garments/s end-to-end predict (PointNet++ -> gridding -> UNet3D -> WNF decode -> marching cubes)",
            "value": garments / tmax, "unit": "garments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_mode == "fp32" else f"f32 ({args.conv_mode} operand split on the 16-bit matrix cores for the 3x3x3 convs, fp32 accumulation; everything else fp32/fp64)",
            "data": "synthetic",
            "config": {"workload": f"full conv_implicit_wnf pipeline, batch={args.batch}/GPU, {args.points}-pt clouds, "
                                   f"{args.grid}^3 feature volume ({args.reduce}), {args.volume_size}^3 WNF + GGM + MC33 + surface decode",
                       "batch_per_gpu": args.batch, "points": args.points, "grid": args.grid, "reduce": args.reduce,
                       "volume_size": args.volume_size, "iso_level": "mid(min,max)" if auto_level else 0.5,
                       "weights": "seeded synthetic (reference architecture)", "mesh_verts_per_step": verts_total,
                       "parallelism": f"dp{world} (independent garment shards, no data-path collective)"},
            "stages_ms": stages_ms,
            "roofline": {"bound": "mfma", "kernel": key, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_note": peak_note,
                         "traffic": measured_traffic(args, key),
                         "launches": g["n"], "avg_launch_ms": g["ms"] / g["n"], "flops_per_launch": g["flops"] / g["n"],
                         "algorithmic_bytes_per_launch": g["bytes"] / g["n"],
                         "hbm_frac_of_8TBs": g["bytes"] / (g["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "all_conv_instances": {k: {"launches": v["n"], "ms": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12}
                                                for k, v in groups.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, hp, sd)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
