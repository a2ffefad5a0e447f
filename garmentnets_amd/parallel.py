"""Multi-GPU plumbing: one process per GPU, garments sharded contiguously, NO data-path collective.

The path shards by garment: every op is segmented by example and the 30 MB of weights are replicated, so
garment batches never move between GPUs.  The only collective is an all-gather of a small fixed-size metrics vector per
rank (RCCL over xGMI on GPUs, gloo in the CPU tests) -- latency-bound, bandwidth irrelevant.
"""
import os

import torch
import torch.distributed as dist

METRIC_SLOTS = 8   # [garments, seconds, + 6 spare per-stage slots]


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL on ROCm
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` garments owned by `rank`; remainders go to the lowest ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(global_batch, n_points, seed, rank, world, colour="uniform"):
    """this rank's contiguous shard of the seeded GLOBAL batch of synthetic garments (BASELINE config[3]: 128 garments over 8 GPUs):
    -> (Batch on the host with batch ids restarting at 0, (lo, hi)).  Garment g of the global batch depends on (seed, g) only, so the
    concatenation of the shards over the ranks IS synthetic_cloud(global_batch, n_points, seed) -- tests/test_parallel_cpu.py."""
    from . import synthetic
    from .batch import Batch
    lo, hi = shard_range(global_batch, rank, world)
    x, pos, batch = synthetic.synthetic_cloud(hi - lo, n_points, seed=seed, first=lo, colour=colour)
    return Batch(sizes=[n_points] * (hi - lo), x=x, pos=pos, batch=batch), (lo, hi)


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def local_cpus_of_device(index):
    """host cores on the NUMA node of HIP device `index` (sysfs local_cpulist of its PCI function) or None when sysfs does not tell"""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        import glob
        for d in glob.glob(f"/sys/bus/pci/devices/{bdf}.*"):
            cpus = _parse_cpulist(open(os.path.join(d, "local_cpulist")).read())
            if cpus:
                return cpus
    except Exception:
        pass
    return None


def plan_affinity(local_rank, local_world, device_of_rank, cpus_of_device, allowed):
    """pure planning (tested on CPU): the cores rank `local_rank` of `local_world` local ranks should run on.  Every rank takes its
    device's NUMA-local cores (restricted to `allowed`, the process's current mask); ranks whose devices share a core list split it into
    equal contiguous slices in rank order, so that eight host-side tails (mesh slicing, pinned D2H, Python) never pile onto one node's
    cores.  -> sorted list of cores (never empty)"""
    allowed = sorted(allowed)
    lists = []
    for r in range(local_world):
        cpus = cpus_of_device(device_of_rank(r))
        cpus = [c for c in (cpus or allowed) if c in set(allowed)] or allowed
        lists.append(tuple(cpus))
    mine = lists[local_rank]
    peers = [r for r in range(local_world) if lists[r] == mine]
    k, n = peers.index(local_rank), len(peers)
    per = max(1, len(mine) // n)
    sl = list(mine[k * per:(k + 1) * per]) if k * per < len(mine) else [mine[k % len(mine)]]
    return sl or list(mine)


def pin_rank(local_rank, local_world, device_index, max_threads=8):
    """bind this process to its GPU's NUMA-local share of the host cores (plan_affinity) and size torch's CPU thread pool to it.
    -> description dict for the bench line; never raises (a container without sched_setaffinity keeps its mask)"""
    info = {"pinned": False}
    try:
        allowed = os.sched_getaffinity(0)
        ndev = max(1, torch.cuda.device_count())
        cores = plan_affinity(local_rank, local_world, lambda r: r % ndev, local_cpus_of_device, allowed)
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(max_threads, len(cores))))
        info.update(pinned=True, cores=len(cores), first_core=cores[0], last_core=cores[-1], torch_threads=torch.get_num_threads(),
                    numa_local=local_cpus_of_device(device_index) is not None)
    except Exception as e:      # noqa: BLE001
        info["why_not"] = repr(e)
    return info


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_metrics(values, device="cpu"):
    """all-gather a short list of floats from every rank -> list (per rank) of lists."""
    v = list(values) + [0.0] * (METRIC_SLOTS - len(values))
    assert len(v) == METRIC_SLOTS
    t = torch.tensor(v, dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def gather_vector(values, device="cpu"):
    """all-gather an equal-length list of floats from every rank -> list (per rank) of lists (bench.py: the per-garment result
    checksums of every shard, so that rank 0 can print them in global garment order)"""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def aggregate_throughput(per_rank):
    """per_rank: [[garments, seconds, ...], ...] -> (whole-job garments/s with the MAX time over ranks, max seconds)."""
    garments = sum(r[0] for r in per_rank)
    tmax = max(r[1] for r in per_rank)
    return garments / tmax, tmax
