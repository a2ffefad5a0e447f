#!/bin/bash
# Profiles of the default bench command for one round (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/):
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel time
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE       -> HBM traffic (separate passes: the TCC block has 4 slots, FETCH_SIZE needs 3)
#   3. --pmc SQ_* matrix-core / LDS counters     -> MFMA utilisation, LDS conflicts (north_star: "MFMA utilisation reported")
# PMC passes carry --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
RAW=/tmp/prof_$TAG                      # raw rocprofv3 output stays on the box (it exceeds gpurun's 64 MiB merge-back limit)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$RAW"
cd /tmp && export TMPDIR=/tmp
exec < /dev/null                            # nothing below may wait on stdin
# one stream: counter collection serialises kernels anyway, and per-kernel rows are easier to read
export GARMENTNETS_PREFETCH_ZERO=0
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict-pass --no-host-io-pass --no-validate --no-occupancy-pass --no-in-flight-pass --no-latency-b1 --no-pmc $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$RAW/fetch" -o pmc -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$RAW/write" -o pmc -- $BENCH > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$RAW/mfma" -o pmc -- $BENCH > "$OUT/mfma.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES \
    -d "$RAW/issue" -o pmc -- $BENCH > "$OUT/issue.log" 2>&1
# summaries only travel back
cd "$REPO"
find "$RAW" -name "*.csv" -printf "%p %s\n" > "$OUT/files.txt"
K=$(find "$RAW/stats" -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp "$K" "$OUT/kernel_stats.csv"
F=$(dirname "$(find "$RAW/fetch" -name "*counter_collection.csv" | head -1)" < /dev/null); W=$(dirname "$(find "$RAW/write" -name "*counter_collection.csv" | head -1)" < /dev/null)
python tools/pmc_summary.py "$F" "$W" 3 > "$OUT/hbm_traffic.json" 2> "$OUT/hbm_traffic.err"
python tools/pmc_mfma_summary.py "$RAW" > "$OUT/mfma_util.json" 2> "$OUT/mfma_util.err"
M=$(find "$RAW/mfma" -name "*counter_collection.csv" | head -1); [ -n "$M" ] && head -3 "$M" > "$OUT/mfma_csv_head.txt"
