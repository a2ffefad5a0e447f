mkdir -p gpurun_out/r05_relu
for a in "256" "256 rest" "256" "256 rest"; do echo "== $a"; timeout 200 python tools/dev/ab_decoder.py $a 2>&1 | grep "M=1048576\|M=16777216"; done 2>&1 | tee gpurun_out/r05_relu/rest.txt
