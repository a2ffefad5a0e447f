#!/bin/bash
# round 6: same-box A/B of the bounded-grid fp32 decoder (gated no-op launches) against the build with HEAD's decode.hip
O=gpurun_out/r06_dec_ab; mkdir -p $O
F="--no-cpu-baseline --no-in-flight-pass --no-pmc --no-occupancy-pass --no-strict-pass --no-host-io-pass --no-validate --steps 10 --warmup 3"
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export GARMENTNETS_HIP_LIB=$PWD/tools/dev/_build/lib_olddec.so; else unset GARMENTNETS_HIP_LIB; fi
    python bench.py $F > $O/$v$i.json 2> $O/$v$i.err
    python - $v$i <<'PY'
import json,sys; d=json.load(open("gpurun_out/bench_detail.json")); print(sys.argv[1], round(d["value"],2), {k[:12]:round(v,2) for k,v in d["stages_ms"].items()}, "b1", round(d["latency_b1"]["ms_median"],3), {k[:8]:round(v,3) for k,v in d["latency_b1"]["stages_ms"].items()})
PY
  done
done
