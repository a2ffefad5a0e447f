#!/bin/bash
# dev-only: ordered kernel timeline of ONE bench step (rocprofv3 --kernel-trace), with the idle gaps between consecutive kernels on the GPU:
# where the step is launch-bound and which small kernels sit between the large ones.   usage: tools/dev/step_timeline.sh <tag> [bench args]
TAG=${1:-tl}; shift || true
REPO=$(cd "$(dirname "$0")/../.." && pwd)
RAW=/tmp/tl_$TAG; OUT=$REPO/gpurun_out/tl_$TAG; mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
export GARMENTNETS_PREFETCH_ZERO=${GARMENTNETS_PREFETCH_ZERO:-1}
rocprofv3 --kernel-trace --output-format csv -d $RAW -o tl -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict-pass --no-host-io-pass --no-validate --no-occupancy-pass --no-in-flight-pass --no-latency-b1 --no-pmc $* > $OUT/run.log 2>&1
F=$(find $RAW -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = from the last fps_kernel launch back to ... : take the last occurrence of the first-level fps kernel as the step's start
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("void fps_kernel") or n.startswith("fps_kernel")]
# two fps launches per step (two SA levels): the step starts a little before the second-to-last one
i0 = starts[-2] if len(starts) >= 2 else 0
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
tot_gap = 0.0; busy = 0.0
print("# idx  start_us  dur_us  gap_before_us  kernel")
for i in range(max(0, i0 - 12), len(rows)):
    r = rows[i]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    if i >= i0: tot_gap += max(gap, 0.0); busy += (e - s) / 1e3
    print(f"{i - i0:5d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:9.1f}  {r['Kernel_Name'][:110]}")
    prev_end = max(prev_end, e)
print(f"# from the step's first fps launch to the trace's end: kernels {busy / 1e3:.2f} ms busy (sum over streams), idle gaps {tot_gap / 1e3:.2f} ms")
PY
wc -l $OUT/timeline.txt; tail -1 $OUT/timeline.txt
