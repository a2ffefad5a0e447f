#!/bin/bash
# round 6: the bench line as the driver runs it, then the two workloads whose dominant kernel is the decoder MLP (gpurun -- bash tools/dev/run_r06_bench.sh)
O=gpurun_out/r06_bench; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default.json 2> $O/default.err; echo "default rc $?"; tail -c 3000 $O/default.json
cp gpurun_out/bench_detail.json $O/default_detail.json
python bench.py --volume-size 256 --batch 8 --steps 5 --no-strict-pass --no-latency-b1 --no-cpu-baseline > $O/q256.json 2> $O/q256.err; echo "q256 rc $?"; tail -c 2500 $O/q256.json
cp gpurun_out/bench_detail.json $O/q256_detail.json
python bench.py --grid 32 --reduce max --steps 10 --no-strict-pass --no-latency-b1 --no-cpu-baseline > $O/g32.json 2> $O/g32.err; echo "g32 rc $?"; tail -c 2500 $O/g32.json
cp gpurun_out/bench_detail.json $O/g32_detail.json
