#!/bin/bash
# The 1/2/4/8-GPU weak-scaling sweep of BASELINE config[3] (16 garments per GPU; 128 over 8 GPUs) exactly as the driver launches bench.py:
#   N=1: python bench.py --gpus 1 ...          N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
# One JSON line per N is appended to gpurun_out/scale/scale.jsonl; every N>1 line carries scaling_vs_n1 (value / (N x the N=1 value of THIS
# sweep)) and the per-rank seconds vector.  usage: tools/run_scale.sh [max_gpus] [bench args...]      (needs that many GPUs on one node)
set -u
MAXN=${1:-8}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/scale
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--steps 10 --warmup 3 --no-strict-pass --no-host-io-pass --no-occupancy-pass --no-in-flight-pass --no-validate --no-cpu-baseline --no-pmc $*"
N1=""
for N in 1 2 4 8; do
    [ "$N" -gt "$MAXN" ] && break
    EXTRA=""; [ -n "$N1" ] && EXTRA="--n1-value $N1"
    if [ "$N" -eq 1 ]; then
        LINE=$(cd "$REPO" && python bench.py --gpus 1 $COMMON | tail -1)
    else
        PORT=$((29500 + N))
        LINE=$(cd "$REPO" && python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
               bench.py --gpus "$N" $COMMON $EXTRA | tail -1)
    fi
    echo "$LINE" >> "$OUT/scale.jsonl"
    V=$(echo "$LINE" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null) || V=""
    [ "$N" -eq 1 ] && N1=$V
    echo "N=$N value=$V garments/s  (n1=$N1)"
done
