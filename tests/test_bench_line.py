"""bench.py's stdout contract: ONE compact JSON line the driver can recover from the tail of stdout (round 5's 21.6 KB line came back as
`parsed: null`).  The canned inputs are full lines earlier rounds committed under profiles/ -- the detail dict bench.py still builds."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


def _full(name):
    with open(os.path.join(REPO, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r05_bench_default.json", "r05_bench_q256.json", "r05_bench_g32.json", "r05_bench_pointnet2.json", "r05_bench_noisy_wnf.json"])
def test_compact_line_is_small_ordered_and_complete(name):
    bench = _bench()
    full = _full(name)
    if name == "r05_bench_default.json":
        assert len(json.dumps(full)) > 16384                 # the canned dict is the very line that broke the reader
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT == 6144
    back = json.loads(text)
    assert tuple(back) == bench.COMPACT_KEYS                 # contract keys first, in this order
    assert tuple(back)[:12] == ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    assert tuple(back["roofline"]) == bench.ROOFLINE_KEYS
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac"):
        assert back["roofline"][k] is not None
    assert back["roofline"]["frac"] == pytest.approx(back["roofline"]["achieved"] / back["roofline"]["peak"], rel=1e-3)
    if full.get("cpu_baseline") is None:                     # (a side run made with --no-cpu-baseline)
        assert back["cpu_baseline"] is None and name != "r05_bench_default.json"
    else:
        assert set(back["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and back["cpu_baseline"]["kind"] in ("port", "reference")
    assert "workload" in back["config"] and "model" not in back["config"]
    assert back["value"] == pytest.approx(full["value"], rel=1e-5) and back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert back["higher_is_better"] is True and back["vs_baseline"] is None and back["detail"] == "gpurun_out/bench_detail.json"
    # no nested prose, no hwmon / throttle dicts: every leaf is a scalar or a short string
    def leaves(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from leaves(v)
        elif isinstance(o, list):
            for v in o:
                yield from leaves(v)
        else:
            yield o
    assert all(not isinstance(v, str) or len(v) <= 200 for v in leaves(back))


def test_compact_line_with_eight_ranks_still_fits():
    bench = _bench()
    full = _full("r05_bench_default.json")
    full.update(n_gpus=8, rccl_ranks_seen=8, dist_backend="nccl", scaling_vs_n1=0.97123456)
    full["per_rank"] = {"seconds": [2.1] * 8, "garments_per_s": [150.123456 + i for i in range(8)], "slowest_over_fastest": 1.0123456, "host_affinity_rank0": {"pinned": True}}
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
    assert line["rccl_ranks_seen"] == 8 and line["dist_backend"] == "nccl" and len(line["per_rank"]["garments_per_s"]) == 8
    assert line["scaling_vs_n1"] == pytest.approx(0.97123, abs=1e-5)


def test_pick_roofline_names_the_group_with_the_most_time():
    bench = _bench()
    g = lambda ms, work, n=4, byts=1e9: dict(ms=ms, work=work, bytes=byts, n=n)
    groups = {"conv3d_split_wino_kernel<true>": g(40.0, 3e13), "conv3d_split_strip_kernel<true>": g(30.0, 1.3e13),
              "implicit_decode_split_kernel<1, 2, false>": g(55.0, 2.3e13, n=128), "fps_kernel": g(2.0, 1e8), "trilinear_brick_kernel": g(5.0, 0.0)}
    args = type("A", (), {})()
    rl = bench.pick_roofline(args, groups, "f16x2", "f16x2")
    assert rl["kernel"] == "implicit_decode_split_kernel<1, 2, false>" and rl["bound"] == "mfma"
    assert rl["peak"] == pytest.approx(2500.0 / 3) and rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"])
    assert rl["share_of_bracketed_ms"] == pytest.approx(55.0 / 132.0)
    groups["implicit_decode_split_kernel<1, 2, false>"]["ms"] = 10.0
    rl = bench.pick_roofline(args, groups, "f16x2", "f16x2")
    assert rl["kernel"] == "conv3d_split_wino_kernel<true>" and rl["peak"] == pytest.approx(2500.0 / 2)       # 3 x 36/54 executed products
    assert set(rl["all_conv_instances"]) == {"conv3d_split_wino_kernel<true>", "conv3d_split_strip_kernel<true>"}
