"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py (garment sharding + the metrics all-gather)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from garmentnets_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = parallel.init(backend="gloo")
    lo, hi = parallel.shard_range(total, r, w)
    parallel.barrier()
    per_rank = parallel.gather_metrics([hi - lo, 1.0 + 0.5 * r], device="cpu")      # rank 1 is the slow one
    value, tmax = parallel.aggregate_throughput(per_rank)
    q.put((r, lo, hi, per_rank, value, tmax))
    torch.distributed.destroy_process_group()


def test_shard_ranges_cover_and_balance():
    for total in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_range(128, 3, 8) == (48, 64)       # BASELINE config 4: 16 garments per GPU


def test_two_rank_gloo_metrics_gather():
    world, total = 2, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, pr0, v0, t0), (r1, lo1, hi1, pr1, v1, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 17, 17, 33)
    assert pr0 == pr1 and [p[0] for p in pr0] == [17.0, 16.0] and [p[1] for p in pr0] == [1.0, 1.5]
    assert v0 == v1 == 33 / 1.5 and t0 == 1.5        # whole-job garments / MAX time over ranks


def test_single_process_passthrough():
    assert parallel.gather_metrics([4, 2.0]) == [[4.0, 2.0] + [0.0] * 6]
    assert parallel.aggregate_throughput([[4, 2.0]]) == (2.0, 2.0)


def _shard_worker(rank, world, port, total, n, seed, q):
    import zlib
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = parallel.init(backend="gloo")
    shard, (lo, hi) = parallel.shard_batch(total, n, seed, r, w)
    # what the data path consumes on this rank: per-garment checksums of its shard, exchanged through the metrics all-gather
    sums = [float(zlib.crc32(shard.pos[i * n:(i + 1) * n].numpy().tobytes()) % 65521) for i in range(hi - lo)]
    gathered = parallel.gather_metrics([hi - lo] + sums, device="cpu")
    q.put((r, lo, hi, gathered, int(shard.batch.max()) if hi > lo else -1))
    torch.distributed.destroy_process_group()


def test_two_rank_global_batch_is_the_concatenation_of_the_shards():
    """config[3] in miniature: a seeded global batch of 5 garments, sharded over 2 gloo ranks with shard_batch; the shards, in rank order,
    are exactly the single-process global batch (garment by garment), with local batch ids restarting at 0 on every rank"""
    import zlib
    from garmentnets_amd import synthetic as S
    world, total, n, seed = 2, 5, 64, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, total, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, gpos, _ = S.synthetic_cloud(total, n, seed=seed)
    want = [float(zlib.crc32(gpos[i * n:(i + 1) * n].numpy().tobytes()) % 65521) for i in range(total)]
    (r0, lo0, hi0, g0, m0), (r1, lo1, hi1, g1, m1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5) and (m0, m1) == (2, 1)
    assert g0 == g1 and len(g0) == 2                                   # every rank saw both ranks' records
    got = g0[0][1:1 + int(g0[0][0])] + g0[1][1:1 + int(g0[1][0])]
    assert got == want


def test_plan_affinity_splits_numa_local_cores_between_the_ranks_that_share_them():
    """bench.py pins every rank to its GPU's NUMA-local cores; ranks whose GPUs hang off the same node split that node's cores (8 ranks on a
    2-socket host: 4 + 4), a container mask is respected, and unknown topology falls back to an even split of the allowed cores"""
    from garmentnets_amd import parallel
    node = {0: list(range(0, 64)), 1: list(range(64, 128))}
    cpus_of = lambda d: node[d // 4]                      # GPUs 0-3 on socket 0, 4-7 on socket 1
    allowed = set(range(128))
    plans = [parallel.plan_affinity(r, 8, lambda r: r, cpus_of, allowed) for r in range(8)]
    assert all(len(p) == 16 for p in plans)
    assert plans[0] == list(range(0, 16)) and plans[3] == list(range(48, 64)) and plans[4] == list(range(64, 80))
    assert len(set(c for p in plans for c in p)) == 128      # disjoint cover
    # two ranks sharing ONE device (the gloo test mode) share its node: halves
    two = [parallel.plan_affinity(r, 2, lambda r: 0, cpus_of, allowed) for r in range(2)]
    assert two[0] == list(range(0, 32)) and two[1] == list(range(32, 64))
    # container mask: only cores 10..19 allowed -> NUMA list intersected
    masked = parallel.plan_affinity(1, 2, lambda r: r, lambda d: node[0], set(range(10, 20)))
    assert masked == list(range(15, 20))
    # sysfs silent -> even split of the allowed cores; more ranks than cores -> still one core each
    assert parallel.plan_affinity(1, 4, lambda r: r, lambda d: None, set(range(8))) == [2, 3]
    assert parallel.plan_affinity(5, 8, lambda r: r, lambda d: None, {0, 1, 2}) in ([0], [1], [2])
    assert parallel._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
