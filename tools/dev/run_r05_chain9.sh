Q="--no-in-flight-pass --no-latency-b1 --no-pmc --no-strict-pass --no-host-io-pass --no-occupancy-pass --no-cpu-baseline"
show() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))
"; }
echo "== bench $1 #1"; GARMENTNETS_HIP_LIB=$1 timeout 600 python bench.py $Q 2>/dev/null | show
echo "== bench $1 #2"; GARMENTNETS_HIP_LIB=$1 timeout 600 python bench.py $Q 2>/dev/null | show
