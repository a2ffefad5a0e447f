#!/bin/bash
# round 6: bench A/B of the wave-specialised 32-wide Winograd kernel (GARMENTNETS_WINO32_PC=1) against the shipped one, same box, alternating
F="--no-strict-pass --no-latency-b1 --no-cpu-baseline --no-in-flight-pass --no-pmc --no-host-io-pass --no-validate --steps 10 --warmup 3"
for rep in 1 2; do for pc in 0 1; do
GARMENTNETS_WINO32_PC=$pc python bench.py $F > /dev/null 2> /dev/null; python - $pc <<'PY'
import json,sys; d=json.load(open("gpurun_out/bench_detail.json")); c=d["roofline"]["all_conv_instances"]; l=d["literal_affine"]["roofline"]["all_conv_instances"]
k="conv3d_split_wino32_kernel<true>"
print("pc", sys.argv[1], "value %.2f literal %.2f occupancy %.2f | wino32 ms/step %.2f (literal %.2f)" % (d["value"], d["literal_affine"]["value"], d["occupancy_aware"]["value"], c[k]["ms"]/10, l[k]["ms"]/10))
PY
done; done
