O=gpurun_out/r05_ab; mkdir -p $O; B=tools/dev/_build
timeout 300 python tools/dev/ab_strip.py > $O/strip_prefetch.txt 2>&1
GARMENTNETS_HIP_LIB=$B/lib_strip_base.so timeout 300 python tools/dev/ab_strip.py > $O/strip_base3.txt 2>&1
echo new; grep TF $O/strip_prefetch.txt | cut -c1-120; echo base; grep TF $O/strip_base3.txt | cut -c1-120
timeout 200 python tools/dev/ab_zero.py 2>&1 | grep -v amdgpu | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "conv3d or sparse_first or affine_in_weights or polyphase or unet" 2>&1 | tail -4
