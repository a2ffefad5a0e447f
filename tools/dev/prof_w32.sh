#!/bin/bash
# round 6: counters of the 32-wide Winograd kernel next to the x-strip kernel on the same shapes (gpurun -- bash tools/dev/prof_w32.sh <tag>)
TAG=${1:-w32}; REPO=$(pwd); RAW=/tmp/prof_$TAG; OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
exec < /dev/null
CMD="python $REPO/tools/dev/ab_wino32.py prof"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $RAW/fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $RAW/write -o pmc -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $RAW/mfma -o pmc -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES -d $RAW/issue -o pmc -- $CMD > $OUT/issue.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d $RAW/vmem -o pmc -- $CMD > $OUT/vmem.log 2>&1
cd $REPO
F=$(dirname "$(find $RAW/fetch -name '*counter_collection.csv' | head -1)"); W=$(dirname "$(find $RAW/write -name '*counter_collection.csv' | head -1)")
python tools/pmc_summary.py "$F" "$W" 1 > $OUT/hbm_traffic.json 2> $OUT/hbm_traffic.err
python tools/pmc_mfma_summary.py $RAW > $OUT/mfma_util.json 2> $OUT/mfma_util.err
V=$(find $RAW/vmem -name '*counter_collection.csv' | head -1)
[ -n "$V" ] && python - "$V" > $OUT/vmem.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "conv3d" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, v in agg.items():
    print(k, "launches", n[k], {c: x / n[k] for c, x in v.items()})
PY
for f in $OUT/*.log; do tail -n 2 $f; done
