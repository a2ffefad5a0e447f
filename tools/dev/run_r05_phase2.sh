mkdir -p gpurun_out/r05_relu
for lib in "" tools/dev/_build/lib_phase16.so tools/dev/_build/lib_phase32.so tools/dev/_build/lib_phase64.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 2>&1 | grep "M=1048576\|M=16777216"
done 2>&1 | tee gpurun_out/r05_relu/phase.txt
