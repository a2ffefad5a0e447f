#!/bin/bash
# round 6: the new kernel's tests, the pipeline tests that run through it, and the bench A/B (--winograd32 on / off), one box
O=gpurun_out/r06_a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "winograd or unet or conv" 2>&1 | tail -5
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bench_batch or dense_unet or realistic" 2>&1 | tail -5
F="--no-strict-pass --no-latency-b1 --no-cpu-baseline --no-in-flight-pass --no-pmc --steps 10 --warmup 3"
python bench.py $F --winograd32 off > $O/w32off.json 2> $O/w32off.err; python - <<PY
import json; d=json.load(open("gpurun_out/bench_detail.json")); print("off", d["value"], d["literal_affine"]["value"], {k:(round(v["ms"]/10,2), round(v["tflops"])) for k,v in d["roofline"]["all_conv_instances"].items()})
PY
python bench.py $F --winograd32 on > $O/w32on.json 2> $O/w32on.err; python - <<PY
import json; d=json.load(open("gpurun_out/bench_detail.json")); print("on ", d["value"], d["literal_affine"]["value"], {k:(round(v["ms"]/10,2), round(v["tflops"])) for k,v in d["roofline"]["all_conv_instances"].items()})
PY
