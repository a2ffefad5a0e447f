#!/bin/bash
# dev-only: the two SQ counter passes of tools/profile_round.sh around ANY command (an A/B script instead of bench.py) -> gpurun_out/prof_<tag>/mfma_util.json
# usage: tools/dev/prof_ab.sh <tag> <command...>      (run on the GPU box through gpurun)
set -u
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/../.." && pwd)
RAW=/tmp/prof_$TAG
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$RAW"
cd /tmp && export TMPDIR=/tmp
exec < /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$RAW/mfma" -o pmc -- "$@" > "$OUT/mfma.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES \
    -d "$RAW/issue" -o pmc -- "$@" > "$OUT/issue.log" 2>&1
cd "$REPO"
python tools/pmc_mfma_summary.py "$RAW" > "$OUT/mfma_util.json" 2> "$OUT/mfma_util.err"
python - "$OUT/mfma_util.json" "$RAW" <<'PY'
import json, sys, glob, csv, collections, os
d = json.load(open(sys.argv[1]))
# LDS array occupancy from the issue pass: SQ_LDS_IDX_ACTIVE / (CUs x cycles)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(os.path.join(sys.argv[2], "issue", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, e in d["kernels"].items():
    if not ("conv3d" in k or "decode" in k or "upconv" in k):
        continue
    lds_act = agg.get(k, {}).get("SQ_LDS_IDX_ACTIVE")
    print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in e.items() if kk in ("launches", "mfma_util", "effective_clock_ghz", "duration_ms_per_launch",
          "lds_insts_per_512_mfma_mops", "lds_bank_conflict_cycles_per_lds_inst", "lds_issue_stall_share_of_wave_cycles", "wave_cycles_issuing",
          "wave_cycles_parked_waitcnt_or_barrier", "wave_cycles_issue_stalled")}, "LDS_IDX_ACTIVE", lds_act, "gui", e.get("gui_active_cycles_sum_over_xcds"))
PY
