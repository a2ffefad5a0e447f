#!/bin/bash
# dev-only: round-5 A/B runs on the GPU box (variant libraries built by tools/dev/build_variant.sh; outputs under gpurun_out/r05_ab/)
O=gpurun_out/r05_ab; mkdir -p $O
B=tools/dev/_build
timeout 120 $B/roof_burn > $O/roof_burn.txt 2>&1
timeout 200 python tools/dev/ab_wino.py abl > $O/wino_base.txt 2>&1
for v in noflush noconvert nodma nobarrier norows; do
  [ -f $B/lib_wn_$v.so ] && GARMENTNETS_HIP_LIB=$B/lib_wn_$v.so timeout 200 python tools/dev/ab_wino.py abl > $O/wino_$v.txt 2>&1
done
timeout 300 python tools/dev/ab_strip.py > $O/strip_base.txt 2>&1
[ -f $B/lib_dma_strip.so ] && GARMENTNETS_HIP_LIB=$B/lib_dma_strip.so timeout 300 python tools/dev/ab_strip.py > $O/strip_dma.txt 2>&1
timeout 300 python tools/dev/ab_decoder.py > $O/decoder_base.txt 2>&1
[ -f $B/lib_dma_decoder.so ] && GARMENTNETS_HIP_LIB=$B/lib_dma_decoder.so timeout 300 python tools/dev/ab_decoder.py > $O/decoder_dma.txt 2>&1
grep -h "TF(eq)\|fp64" $O/*.txt | grep -v amdgpu | cut -c1-200
