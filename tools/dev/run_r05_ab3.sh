O=gpurun_out/r05_ab; mkdir -p $O; B=tools/dev/_build
timeout 400 python -m pytest tests/test_gpu_parity.py -q -k "winograd or affine_in_weights or sparse_first" 2>&1 | tail -3
timeout 200 python tools/dev/ab_wino.py time > $O/wino_epi_new.txt 2>&1
GARMENTNETS_HIP_LIB=$B/lib_wn_v2.so timeout 200 python tools/dev/ab_wino.py time > $O/wino_epi_old.txt 2>&1
echo new; grep "wino" $O/wino_epi_new.txt | tail -7 | cut -c1-100; echo old; grep "wino" $O/wino_epi_old.txt | tail -7 | cut -c1-100
timeout 300 python tools/dev/bench_nondefault.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/bench_nondefault.txt
