mkdir -p gpurun_out/r05_relu
for pr in 0 -1 0 -1; do
  echo "== GARMENTNETS_SIDE_STREAM_PRIORITY=$pr"
  GARMENTNETS_SIDE_STREAM_PRIORITY=$pr timeout 300 python bench.py --steps 10 --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-occupancy-pass 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('headline', round(d['value'],1), 'two_in_flight', round(d['two_in_flight']['value'],1), 'with_host_io', round(d['with_host_io']['value'],1), 'two_in_flight_with_host_io', round(d['two_in_flight_with_host_io']['value'],1)); print(d['garment_checksums'][:2])
"
done 2>&1 | tee gpurun_out/r05_relu/prio.txt
