mkdir -p gpurun_out/r05_mix
for rep in 1 2; do
for lib in "" tools/dev/_build/lib_r05_nomix.so; do
  echo "== lib=${lib:-mix}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 2>&1 | grep "M=" 
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 512 2>&1 | grep "M=16777216"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep wino | cut -c1-100
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_zero.py 2>&1 | grep randn | cut -c1-100
  GARMENTNETS_HIP_LIB=$lib timeout 300 python bench.py --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-host-io-pass 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['literal_affine']['value'], d['occupancy_aware']['value'], d['stages_ms']); print({k:(round(v['ms'],1), round(v['tflops'])) for k,v in d['roofline']['all_conv_instances'].items()}); print(d['garment_checksums'][:3])
"
done; done 2>&1 | tee gpurun_out/r05_mix/ab.txt
