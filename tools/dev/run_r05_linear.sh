mkdir -p gpurun_out/r05_relu
for lib in tools/dev/_build/lib_lin16.so "" tools/dev/_build/lib_lin16.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_linear.py 2>&1 | grep "M=\|sum\|Error\|error" 
done 2>&1 | tee gpurun_out/r05_relu/linear2.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "linear or mlp or pipeline_against or create_conv or final_conv" 2>&1 | tail -3 | tee -a gpurun_out/r05_relu/linear2.txt
