mkdir -p gpurun_out/r05_relu
for lib in tools/dev/_build/lib_base.so "" tools/dev/_build/lib_base.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_sampler.py 2>&1 | grep "G="
done 2>&1 | tee gpurun_out/r05_relu/sampler.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "lattice or trilinear or sampler" 2>&1 | tail -3 | tee -a gpurun_out/r05_relu/sampler.txt
