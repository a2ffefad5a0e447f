#!/bin/bash
# dev-only: the round's default bench line with a power trace beside it, then the rocprofv3 passes (tools/profile_round.sh)
mkdir -p gpurun_out
python tools/power_trace.py gpurun_out/r3m_power_bench.csv python bench.py --steps 20 --warmup 5 > gpurun_out/r3m_bench_default.json 2> gpurun_out/r3m_bench_default.err
tail -c 300 gpurun_out/r3m_bench_default.err
bash tools/profile_round.sh r3m > gpurun_out/r3m_profile.log 2>&1
ls gpurun_out/prof_r3m
