for rep in 1 2; do
for lib in "" tools/dev/_build/lib_r05_chain_nt.so; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep -v amdgpu | grep wino
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | grep "wino"
done; done 2>&1 | tee gpurun_out/r05_chain/ntstore.txt
