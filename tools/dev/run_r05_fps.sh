mkdir -p gpurun_out/r05_relu
for lib in tools/dev/_build/lib_base.so tools/dev/_build/lib_fpsmid1024.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_fps.py 2>&1 | grep "B="
done 2>&1 | tee gpurun_out/r05_relu/fps3.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "fps or ball or sa_ or pointnet or points or degenerate" 2>&1 | tail -4 | tee -a gpurun_out/r05_relu/fps3.txt
