set -x
mkdir -p gpurun_out/r05_chain
for c in 1 2 3 16; do GARMENTNETS_WINO_CHAIN=$c timeout 120 python tools/dev/wino_chain_check.py 2>&1 | grep -v amdgpu > gpurun_out/r05_chain/check_$c.txt; done
cat gpurun_out/r05_chain/check_*.txt
timeout 300 python tools/dev/ab_wino.py check 2>&1 | grep -v amdgpu | tee gpurun_out/r05_chain/ab_check.txt
timeout 300 python tools/dev/wino_intercept.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_chain/intercept.txt
for c in 1 4 8 32; do echo chain $c; GARMENTNETS_WINO_CHAIN=$c timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep -v amdgpu | tee gpurun_out/r05_chain/abl_$c.txt; done
timeout 900 python -m pytest tests -m gpu -x -q -k "wino or sparse_first or affine_in_weights or graph or sharded or unet" 2>&1 | tail -5
