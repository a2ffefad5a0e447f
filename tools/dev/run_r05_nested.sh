mkdir -p gpurun_out/r05_relu
timeout 600 python -m pytest tests -m gpu -x -q -k "fps or sa_module or pipeline_against or self_loop or sa_fused or smoke or bench_batch" 2>&1 | tail -3 | tee gpurun_out/r05_relu/nested.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05c_bench_default.json 2> gpurun_out/r05c_bench_default.err
python bench.py --steps 10 --warmup 3 --grid 32 --reduce max --no-strict-pass --no-cpu-baseline --no-pmc --no-latency-b1 > gpurun_out/r05c_bench_g32.json 2> gpurun_out/r05c_bench_g32.err
python bench.py --steps 10 --warmup 3 --workload pointnet2 --no-pmc > gpurun_out/r05c_bench_pointnet2.json 2> gpurun_out/r05c_bench_pointnet2.err
tail -c 600 gpurun_out/r05c_bench_default.json
