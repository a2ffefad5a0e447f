#!/bin/bash
# round 6: the fp32 decoder kernel on a bounded grid (gated no-op launches): its tests, the pipeline tests through it, and the step's stage times
O=gpurun_out/r06_dec; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or trilinear or lattice or pipeline or strict or fp32" 2>&1 | tail -4
python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu -k "predict or bench_batch or nan or fallback or q256 or config4" 2>&1 | tail -4
F="--no-cpu-baseline --no-in-flight-pass --no-pmc --no-occupancy-pass --steps 10 --warmup 3"
python bench.py $F > $O/default.json 2> $O/default.err; echo "rc $?"; cp gpurun_out/bench_detail.json $O/default_detail.json
python - <<'PY'
import json; d=json.load(open("gpurun_out/bench_detail.json")); print(d["value"], d["stages_ms"], "strict", d["strict_fp32"]["value"], "b1", d["latency_b1"]["ms_median"], d["latency_b1"]["stages_ms"])
PY
