# dev-only: timing-only ablations of the decoder MLP kernel (wrong results on purpose; A/B record section 22)
mkdir -p gpurun_out/r05_relu
for lib in "" tools/dev/_build/lib_abl_nobar.so tools/dev/_build/lib_abl_noepi.so tools/dev/_build/lib_abl_nofrag.so tools/dev/_build/lib_abl_nodmaloop.so tools/dev/_build/lib_abl_occ1.so tools/dev/_build/lib_abl_nobar_nofrag.so tools/dev/_build/lib_abl_all.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 2>&1 | grep "M=1048576\|M=16777216"
done 2>&1 | tee gpurun_out/r05_relu/decabl.txt
