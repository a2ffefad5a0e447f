#!/bin/bash
# round 6: the numbers of record with the final code: the bench line as the driver runs it + the side workloads + the rocprofv3 passes (gpurun -- bash tools/dev/run_r06_final.sh)
O=gpurun_out/r06_final; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default.json 2> $O/default.err; echo "default rc $?"; cp gpurun_out/bench_detail.json $O/default_detail.json
python bench.py > $O/noflags.json 2> $O/noflags.err; echo "noflags rc $?"; cp gpurun_out/bench_detail.json $O/noflags_detail.json
S="--no-strict-pass --no-latency-b1 --no-cpu-baseline"
python bench.py --volume-size 256 --batch 8 --steps 5 $S > $O/q256.json 2> $O/q256.err; cp gpurun_out/bench_detail.json $O/q256_detail.json
python bench.py --grid 32 --reduce max --steps 10 $S > $O/g32.json 2> $O/g32.err; cp gpurun_out/bench_detail.json $O/g32_detail.json
python bench.py --workload pointnet2 --steps 10 > $O/pointnet2.json 2> $O/pointnet2.err; cp gpurun_out/bench_detail.json $O/pointnet2_detail.json
python bench.py --input noisy_wnf --steps 5 $S > $O/noisy_wnf.json 2> $O/noisy_wnf.err; cp gpurun_out/bench_detail.json $O/noisy_wnf_detail.json
for f in default noflags q256 g32 pointnet2 noisy_wnf; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], len(json.dumps(d)), "B: value", d["value"], "ms", d["ms_per_step"], "| roofline", r["kernel"], r["frac"], "| host_io", (d.get("with_host_io") or {}).get("value"), "| literal", (d.get("literal_affine") or {}).get("value"), "| b1", d.get("latency_b1_ms"))
PY
done
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; echo "profile rc $?"
