mkdir -p gpurun_out/r05_relu
for lib in tools/dev/_build/lib_uporig.so tools/dev/_build/lib_upnochain.so tools/dev/_build/lib_upchain4.so tools/dev/_build/lib_upb3.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_upconv.py 2>&1 | grep "B=\|rror"
done 2>&1 | tee gpurun_out/r05_relu/upchain.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "polyphase or upconv or unet_against or full_size_unet or decoder_conv" 2>&1 | tail -3 | tee -a gpurun_out/r05_relu/upchain.txt
