#!/bin/bash
# round 6: the batch's surface queries in three launches: tests, then same-box bench A/B (GARMENTNETS_DECODE_BATCH_BYTES=0 -> the garment loop)
O=gpurun_out/r06_batchdec; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or decode or pipeline or trilinear or lattice" 2>&1 | tail -4
python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu -k "predict or bench_batch or nan or fallback or slot or config4" 2>&1 | tail -4
F="--no-cpu-baseline --no-in-flight-pass --no-pmc --no-occupancy-pass --no-strict-pass --no-host-io-pass --no-validate --no-latency-b1 --steps 10 --warmup 3"
for i in 1 2; do
  for v in batch loop; do
    if [ $v = loop ]; then export GARMENTNETS_DECODE_BATCH_BYTES=0; else unset GARMENTNETS_DECODE_BATCH_BYTES; fi
    python bench.py $F > $O/$v$i.json 2> $O/$v$i.err
    python - $v$i <<'PY'
import json,sys; d=json.load(open("gpurun_out/bench_detail.json")); print(sys.argv[1], round(d["value"],2), round(d["ms_per_step"],2), {k[:12]:round(v,2) for k,v in d["stages_ms"].items()})
PY
  done
done
