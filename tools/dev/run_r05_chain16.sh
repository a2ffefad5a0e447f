mkdir -p gpurun_out/r05_chain
for c in 1 3 16; do GARMENTNETS_WINO_CHAIN=$c timeout 120 python tools/dev/wino_chain_check.py 2>&1 | grep -v amdgpu | cut -d' ' -f1-8 > gpurun_out/r05_chain/check_$c.txt; done
GARMENTNETS_HIP_LIB=tools/dev/_build/lib_r05_unchained.so timeout 120 python tools/dev/wino_chain_check.py 2>&1 | grep -v amdgpu | cut -d' ' -f1-8 > gpurun_out/r05_chain/check_old.txt
cat gpurun_out/r05_chain/check_1.txt; for c in 3 16 old; do diff gpurun_out/r05_chain/check_1.txt gpurun_out/r05_chain/check_$c.txt > /dev/null && echo "same as $c"; done
for rep in 1 2; do
for lib in "" tools/dev/_build/lib_r05_chain_p1.so tools/dev/_build/lib_r05_unchained.so; do
  echo "== lib=${lib:-new}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep -v amdgpu | grep wino
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | grep "wino"
done; done 2>&1 | tee gpurun_out/r05_chain/p2.txt
