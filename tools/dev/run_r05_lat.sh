# dev-only: the fused lattice sampler (Arith.fused_lattice) against the default (brick sampler + decoder) after the round's decoder changes; new FPS tests
mkdir -p gpurun_out/r05_relu
timeout 300 python -m pytest tests -m gpu -x -q -k "fps" 2>&1 | tail -3 | tee gpurun_out/r05_relu/lat.txt
for fl in 0 1; do
  echo "== GARMENTNETS_FUSED_LATTICE=$fl"
  GARMENTNETS_FUSED_LATTICE=$fl timeout 300 python bench.py --steps 10 --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-host-io-pass 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['literal_affine']['value'], d['stages_ms']); print(d['garment_checksums'][:3])
"
done 2>&1 | tee -a gpurun_out/r05_relu/lat.txt
