#!/bin/bash
# round 6, second half: the gpurun sessions behind sections 9 - 11 of profiles/r06_ab_experiments.txt, one case each (gpurun -- bash tools/dev/run_r06_ab.sh <case>)
#   tail      the tail fold's tests + the tests that run through IsoBatchJob + a short bench (stage times, hbm_members, ggm_accumulation)
#   fps       farthest-point sampling: parity tests, then tools/dev/ab_fps.py (a variant library through GARMENTNETS_HIP_LIB: build_variant.sh / a patch of experiments/)
#   ggm       tools/dev/ab_ggm.py + the scipy goldens for every library tools/dev/_build/lib_ggm_*.so (hipcc -DGGM_TZ=.. -DGGM_TY=.. on csrc/iso.hip)
#   dec       same-box bench A/B of two libraries: the tree's and tools/dev/_build/lib_old.so
#   batchdec  same-box bench A/B of the batched surface queries against the garment loop (GARMENTNETS_DECODE_BATCH_BYTES=0)
CASE=${1:-tail}; O=gpurun_out/r06_$CASE; mkdir -p $O
SHORT="--no-cpu-baseline --no-in-flight-pass --no-pmc --no-occupancy-pass --no-strict-pass --no-host-io-pass --no-validate --steps 10 --warmup 3"
line() { python - "$1" <<'PY'
import json,sys; d=json.load(open("gpurun_out/bench_detail.json")); b1=d.get("latency_b1") or {}
print(sys.argv[1], round(d["value"],2), round(d["ms_per_step"],2), {k[:12]:round(v,2) for k,v in d["stages_ms"].items()}, "b1", b1.get("ms_median"))
PY
}
case $CASE in
tail)
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ggm or iso or batched or marching or predict" 2>&1 | tail -3
  python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu -k "predict or bench_batch or nan or fallback or shell" 2>&1 | tail -3
  python bench.py $SHORT > $O/default.json 2> $O/default.err; cp gpurun_out/bench_detail.json $O/default_detail.json; line tail
  python - <<'PY'
import json; h=json.load(open("gpurun_out/bench_detail.json"))["hbm_members"]; print({k:round(v["ms"],3) for k,v in h.items() if isinstance(v,dict) and "ms" in v}); print(h.get("ggm_accumulation"))
PY
  ;;
fps)
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps or sa_module or ball_query" 2>&1 | tail -3
  python tools/dev/ab_fps.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_tree.txt
  for l in tools/dev/_build/lib_fps_*.so; do [ -f "$l" ] && { echo "--- $l"; GARMENTNETS_HIP_LIB=$PWD/$l python tools/dev/ab_fps.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_$(basename $l .so).txt; }; done
  ;;
ggm)
  for l in garmentnets_amd/libgarmentnets_hip.so tools/dev/_build/lib_ggm_*.so; do [ -f "$l" ] || continue
    echo "--- $l"; GARMENTNETS_HIP_LIB=$PWD/$l python tools/dev/ab_ggm.py 2>&1 | grep -v amdgpu.ids
    GARMENTNETS_HIP_LIB=$PWD/$l python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "isosurface or ggm or shell or batched_iso" 2>&1 | tail -1
  done ;;
dec)
  for i in 1 2; do for v in new old; do
    if [ $v = old ]; then export GARMENTNETS_HIP_LIB=$PWD/tools/dev/_build/lib_old.so; else unset GARMENTNETS_HIP_LIB; fi
    python bench.py $SHORT > $O/$v$i.json 2> $O/$v$i.err; line $v$i
  done; done ;;
batchdec)
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or decode or pipeline or trilinear or lattice" 2>&1 | tail -3
  for i in 1 2; do for v in batch loop; do
    if [ $v = loop ]; then export GARMENTNETS_DECODE_BATCH_BYTES=0; else unset GARMENTNETS_DECODE_BATCH_BYTES; fi
    python bench.py $SHORT --no-latency-b1 > $O/$v$i.json 2> $O/$v$i.err; line $v$i
  done; done ;;
esac
