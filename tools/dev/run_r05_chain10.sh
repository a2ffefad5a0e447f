mkdir -p gpurun_out/r05_chain
Q="--no-in-flight-pass --no-latency-b1 --no-pmc $2"
echo "== bench lib=$1 flags=$Q #1"; GARMENTNETS_HIP_LIB=$1 timeout 600 python bench.py $Q 2>gpurun_out/r05_chain/err1.txt > gpurun_out/r05_chain/out1.json
echo "== #2"; GARMENTNETS_HIP_LIB=$1 timeout 600 python bench.py $Q 2>gpurun_out/r05_chain/err2.txt > gpurun_out/r05_chain/out2.json
for f in gpurun_out/r05_chain/out1.json gpurun_out/r05_chain/out2.json; do python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))
        for k in ('literal_affine','occupancy_aware','strict','host_io','secondary'):
            if k in d: print('  ',k, json.dumps(d[k])[:200])
PY
done
grep -v amdgpu gpurun_out/r05_chain/err1.txt | tail -5
