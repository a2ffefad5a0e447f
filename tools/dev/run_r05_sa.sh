mkdir -p gpurun_out/r05_relu
for lib in "" tools/dev/_build/lib_saA.so tools/dev/_build/lib_saB.so tools/dev/_build/lib_saC.so ""; do
  echo "== lib=${lib:-shipped}"
  GARMENTNETS_HIP_LIB=$lib timeout 200 python bench.py --steps 10 --warmup 3 --workload pointnet2 --batch 16 --no-pmc 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],3)); po=d['roofline']['per_operator']; print({k:round(v['ms'],3) for k,v in po.items()})
"
done 2>&1 | tee gpurun_out/r05_relu/sa.txt
