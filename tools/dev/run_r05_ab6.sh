O=gpurun_out/r05_ab; mkdir -p $O; B=tools/dev/_build
timeout 300 python tools/dev/ab_strip.py > $O/strip_pre.txt 2>&1
GARMENTNETS_HIP_LIB=$B/lib_strip_base.so timeout 300 python tools/dev/ab_strip.py > $O/strip_base4.txt 2>&1
timeout 200 python tools/dev/ab_zero.py > $O/strip_pre_zero.txt 2>&1
GARMENTNETS_HIP_LIB=$B/lib_strip_base.so timeout 200 python tools/dev/ab_zero.py > $O/strip_base4_zero.txt 2>&1
echo new; grep TF $O/strip_pre.txt | cut -c1-120; echo base; grep TF $O/strip_base4.txt | cut -c1-120
echo new; grep TF $O/strip_pre_zero.txt | cut -c1-100; echo base; grep TF $O/strip_base4_zero.txt | cut -c1-100
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "conv3d or sparse_first or affine_in_weights or polyphase or unet" 2>&1 | tail -3
