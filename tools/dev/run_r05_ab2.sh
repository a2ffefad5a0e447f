O=gpurun_out/r05_ab; mkdir -p $O; B=tools/dev/_build
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "hip_graph or sparse_first or winograd" 2>&1 | tail -4
for v in direct03 flush1of4; do GARMENTNETS_HIP_LIB=$B/lib_wn_$v.so timeout 200 python tools/dev/ab_wino.py abl > $O/wino_$v.txt 2>&1; done
GARMENTNETS_HIP_LIB=$B/lib_wn_direct03.so timeout 200 python tools/dev/ab_wino.py check > $O/wino_direct03_check.txt 2>&1
timeout 200 python tools/dev/ab_wino.py abl > $O/wino_base2.txt 2>&1
grep -h "wino   at-rest\|wino at-rest" $O/wino_base2.txt $O/wino_direct03.txt $O/wino_flush1of4.txt $O/wino_direct03_check.txt | cut -c1-250
