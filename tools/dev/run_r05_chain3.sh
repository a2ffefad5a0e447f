mkdir -p gpurun_out/r05_chain
for c in 1 3 16; do GARMENTNETS_WINO_CHAIN=$c timeout 120 python tools/dev/wino_chain_check.py 2>&1 | grep -v amdgpu > gpurun_out/r05_chain/check_$c.txt; done
cat gpurun_out/r05_chain/check_1.txt; diff gpurun_out/r05_chain/check_1.txt gpurun_out/r05_chain/check_3.txt && diff gpurun_out/r05_chain/check_1.txt gpurun_out/r05_chain/check_16.txt && echo CHAIN-INVARIANT
GARMENTNETS_HIP_LIB=tools/dev/_build/lib_r05_unchained.so timeout 120 python tools/dev/wino_chain_check.py 2>&1 | grep -v amdgpu > gpurun_out/r05_chain/check_old.txt; diff gpurun_out/r05_chain/check_1.txt gpurun_out/r05_chain/check_old.txt && echo SAME-AS-UNCHAINED
for rep in 1 2; do
for lib in "" tools/dev/_build/lib_r05_unchained.so; do
  echo "== lib=${lib:-chained}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep -v amdgpu | grep wino
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | grep wino
done; done 2>&1 | tee gpurun_out/r05_chain/ab_same_box2.txt
