mkdir -p gpurun_out/r05_chain
for rep in 1 2; do
for cfg in "1:" "16:" "0:tools/dev/_build/lib_r05_unchained.so"; do
  c=${cfg%%:*}; lib=${cfg#*:}
  echo "== chain=$c lib=${lib:-chained}"
  GARMENTNETS_WINO_CHAIN=$c GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | grep wino
done; done 2>&1 | tee gpurun_out/r05_chain/slope.txt
