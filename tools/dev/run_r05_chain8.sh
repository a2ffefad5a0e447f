mkdir -p gpurun_out/r05_chain
Q="--no-in-flight-pass --no-latency-b1 --no-pmc --no-strict-pass --no-host-io-pass --no-occupancy-pass --no-cpu-baseline"
show() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))
"; }
echo "== first: ab_wino abl (chained)"; timeout 200 python tools/dev/ab_wino.py abl 2>&1 | grep wino
echo "== bench chained #1"; timeout 600 python bench.py $Q 2>gpurun_out/r05_chain/first_err.txt | show
echo "== bench chained #2"; timeout 600 python bench.py $Q 2>/dev/null | show
tail -5 gpurun_out/r05_chain/first_err.txt
