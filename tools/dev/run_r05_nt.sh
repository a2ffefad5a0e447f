mkdir -p gpurun_out/r05_chain
for rep in 1 2; do
for lib in "" tools/dev/_build/lib_r05_no_nt.so; do
  echo "== lib=${lib:-nt}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_zero.py 2>&1 | grep TF | cut -c1-100
  GARMENTNETS_HIP_LIB=$lib timeout 300 python bench.py --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-host-io-pass 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['literal_affine']['value'], d['occupancy_aware']['value']); print({k:(round(v['ms'],1), round(v['tflops'])) for k,v in d['roofline']['all_conv_instances'].items()})
"
done; done 2>&1 | tee gpurun_out/r05_chain/nt_ab.txt
