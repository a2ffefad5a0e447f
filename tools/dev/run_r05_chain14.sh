mkdir -p gpurun_out/r05_chain
for lib in "" tools/dev/_build/lib_r05_unchained.so "" tools/dev/_build/lib_r05_unchained.so; do
echo "== lib=${lib:-chained}"
GARMENTNETS_HIP_LIB=$lib GARMENTNETS_BENCH_STALL_TRACE=1 timeout 600 python bench.py --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate > gpurun_out/r05_chain/st_out.json 2> gpurun_out/r05_chain/st_err.txt
python - <<'PY'
import json
for l in open('gpurun_out/r05_chain/st_out.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['literal_affine']['value'], d['occupancy_aware']['value'])
PY
grep "stall-trace" gpurun_out/r05_chain/st_err.txt | head -2
done
