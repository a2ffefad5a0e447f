#!/bin/bash
# The 1/2/4/8-GPU weak-scaling sweep of BASELINE config[3] (16 garments per GPU; 128 over 8 GPUs) exactly as the driver launches bench.py:
#   N=1: python bench.py --gpus 1 ...          N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
# One JSON line per N goes to profiles/<tag>_scale_n<N>.json (and is appended to gpurun_out/scale/scale.jsonl); every N>1 line carries scaling_vs_n1
# (value / (N x the N=1 value of THIS sweep)) and the per-rank seconds vector.  The sweep also CHECKS the data path: garment g of the seeded global
# batch depends on (seed, g) only and rank r owns garments [16 r, 16 r + 16), so the per-garment checksums of the N-GPU line must begin with the
# checksums of the previous (smaller) N's line, garment for garment, bit for bit -- a rank that computed on the wrong device, shard or stream shows up
# here.  Exit code 1 if they do not.      usage: tools/run_scale.sh [max_gpus] [tag] [bench args...]      (needs that many GPUs on one node)
set -u
MAXN=${1:-8}; shift || true
TAG=${1:-r05}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/scale
mkdir -p "$OUT" "$REPO/profiles"
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--steps 10 --warmup 3 --no-strict-pass --no-host-io-pass --no-occupancy-pass --no-in-flight-pass --no-validate --no-cpu-baseline --no-pmc --no-latency-b1 $*"
N1=""; PREV=""; RC=0
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
    CUR="$REPO/profiles/${TAG}_scale_n${N}.json"
    echo "$LINE" > "$CUR"
    # the per-garment checksums and the per-rank seconds live in the detail file bench.py names in the line (round 6: the stdout line is compact)
    DET="$REPO/profiles/${TAG}_scale_n${N}_detail.json"
    cp "$REPO/gpurun_out/bench_detail.json" "$DET"
    V=$(python -c "import json,sys; print(json.load(open(sys.argv[1]))['value'])" "$CUR" 2>/dev/null) || V=""
    [ "$N" -eq 1 ] && N1=$V
    echo "N=$N value=$V garments/s  (n1=$N1)"
    if [ -n "$PREV" ]; then
        python - "$PREVDET" "$DET" <<'PY' || RC=1
import json, sys
a, b = (json.load(open(p)) for p in sys.argv[1:3])
ca, cb = a["garment_checksums"], b["garment_checksums"]
ok = len(cb) >= len(ca) and cb[:len(ca)] == ca and b.get("rccl_ranks_seen") == b["n_gpus"]
print(f"   garment checksums: the first {len(ca)} of N={b['n_gpus']} {'==' if ok else '!='} N={a['n_gpus']}'s; ranks seen {b.get('rccl_ranks_seen')}, backend {b.get('dist_backend')}")
sys.exit(0 if ok else 1)
PY
    fi
    PREV=$CUR; PREVDET=$DET
done
exit $RC
