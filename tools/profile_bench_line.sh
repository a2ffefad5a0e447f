#!/bin/bash
# The round's default bench line with a 10 Hz power / clock trace beside it, then the rocprofv3 passes (tools/profile_round.sh) and the secondary
# workloads of DESIGN.md section 5.  Run on the GPU box through gpurun; everything lands under gpurun_out/ (copy what is to be judged into profiles/).
# usage: tools/profile_bench_line.sh <tag>
TAG=${1:-r03}
mkdir -p gpurun_out
python tools/power_trace.py gpurun_out/${TAG}_power_bench.csv python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 300 gpurun_out/${TAG}_bench_default.err
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1
python bench.py --steps 10 --warmup 3 --volume-size 256 --batch 8 --no-strict-pass --no-cpu-baseline --no-pmc --no-latency-b1 > gpurun_out/${TAG}_bench_q256.json 2> gpurun_out/${TAG}_bench_q256.err
python bench.py --steps 10 --warmup 3 --input noisy_wnf --no-strict-pass --no-cpu-baseline --no-pmc --no-latency-b1 > gpurun_out/${TAG}_bench_noisy_wnf.json 2> gpurun_out/${TAG}_bench_noisy_wnf.err
python bench.py --steps 10 --warmup 3 --workload pointnet2 --no-pmc > gpurun_out/${TAG}_bench_pointnet2.json 2> gpurun_out/${TAG}_bench_pointnet2.err
python bench.py --steps 10 --warmup 3 --input collapsed --no-strict-pass --no-cpu-baseline --no-pmc --no-latency-b1 > gpurun_out/${TAG}_bench_collapsed.json 2> gpurun_out/${TAG}_bench_collapsed.err
python bench.py --steps 10 --warmup 3 --grid 32 --reduce max --no-strict-pass --no-cpu-baseline --no-pmc --no-latency-b1 > gpurun_out/${TAG}_bench_g32.json 2> gpurun_out/${TAG}_bench_g32.err
ls gpurun_out/prof_${TAG}
