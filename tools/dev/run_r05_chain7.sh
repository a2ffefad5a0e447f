mkdir -p gpurun_out/r05_chain
for lib in "" "" tools/dev/_build/lib_r05_unchained.so ""; do
  echo "== lib=${lib:-chained}"
  GARMENTNETS_HIP_LIB=$lib timeout 600 python bench.py --no-in-flight-pass --no-latency-b1 --no-pmc 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('throttle'))
"
done 2>&1 | tee gpurun_out/r05_chain/bench_ab2.txt
