mkdir -p gpurun_out/r05_chain
for rep in 1 2; do
for st in 0 1 3; do
  echo "== stagger=$st"
  GARMENTNETS_WINO_STAGGER=$st timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep -v amdgpu | grep wino
  GARMENTNETS_WINO_STAGGER=$st timeout 200 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | grep wino
done; done 2>&1 | tee gpurun_out/r05_chain/stagger.txt
