#!/bin/bash
# round 6: the tail fold (range record on the GGM's staging pass, fp32 GGM option): its tests, the tests that run through IsoBatchJob, and the bench's tail stage
O=gpurun_out/r06_tail; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ggm or iso or batched or marching or predict" 2>&1 | tail -5
python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu -k "predict or bench_batch or nan or fallback or shell" 2>&1 | tail -5
F="--no-strict-pass --no-cpu-baseline --no-in-flight-pass --no-pmc --no-occupancy-pass --steps 10 --warmup 3"
python bench.py $F > $O/default.json 2> $O/default.err; echo "rc $?"; cp gpurun_out/bench_detail.json $O/default_detail.json
python - <<'PY'
import json; d=json.load(open("gpurun_out/bench_detail.json")); print(d["value"], d["stages_ms"], d["latency_b1"]["ms_per_step"] if isinstance(d.get("latency_b1"),dict) and "ms_per_step" in d["latency_b1"] else d.get("latency_b1"))
h=d["hbm_members"]; print({k:round(v["ms"],3) for k,v in h.items() if isinstance(v,dict) and "ms" in v}); print(h.get("ggm_accumulation"))
PY
