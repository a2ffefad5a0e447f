#!/bin/bash
# round 6: GGM tile depth (LDS per workgroup -> waves per SIMD) and the fp32 correctly-rounded square root, same box; goldens with every variant
for v in tz4 ty4; do
  export GARMENTNETS_HIP_LIB=$PWD/tools/dev/_build/lib_ggm_$v.so
  echo "--- $v"; python tools/dev/ab_ggm.py 2>&1 | grep -v amdgpu.ids
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "isosurface or ggm or shell or batched_iso" 2>&1 | tail -1
done
