# dev-only A/B: one-instruction ReLU (v_maximum3_f32) + packed bias adds + one-statement stage DMA of the decoder MLPs (A/B record section 21)
mkdir -p gpurun_out/r05_relu
for rep in 1 2; do
for lib in tools/dev/_build/lib_base.so tools/dev/_build/lib_nodma.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 2>&1 | grep "M="
  GARMENTNETS_HIP_LIB=$lib timeout 200 python tools/dev/ab_decoder.py 512 2>&1 | grep "M=16777216"
  [ $rep = 1 ] && GARMENTNETS_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-host-io-pass 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['literal_affine']['value'], d['occupancy_aware']['value'], d['stages_ms']); print({k:(round(v['ms'],1), round(v['tflops'])) for k,v in d['roofline']['all_conv_instances'].items()}); print(d['garment_checksums'][:3])
"
done; done 2>&1 | tee gpurun_out/r05_relu/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or decode or lattice or nan or linear or sa_ or range_contract or smoke" 2>&1 | tail -5 | tee gpurun_out/r05_relu/tests.txt
