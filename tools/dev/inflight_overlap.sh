#!/bin/bash
# dev-only: does batch k+1's front (farthest-point sampling, fused SA, ...) really run beside batch k's UNet with two batches in flight?  rocprofv3 kernel trace of a
# short bench with the in-flight pass; for every fps / sa_fused launch: which kernels overlap it in time.   usage: tools/dev/inflight_overlap.sh <tag> [bench args]
TAG=${1:-ov}; shift || true
REPO=$(cd "$(dirname "$0")/../.." && pwd)
RAW=/tmp/ov_$TAG; OUT=$REPO/gpurun_out/ov_$TAG; mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $RAW -o ov -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict-pass --no-host-io-pass --no-validate --no-occupancy-pass --no-latency-b1 --no-pmc $* > $OUT/run.log 2>&1
F=$(find $RAW -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/overlap.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
t0 = ev[0][0]
big = [e for e in ev if e[1] - e[0] > 200_000]          # kernels longer than 0.2 ms
print("# every fps_kernel<24,...> launch: start ms, duration ms, queue | the long kernels that overlap it (name, overlap ms, queue)")
for s, e, n, q, st in ev:
    if "fps_kernel<24" not in n: continue
    ov = [(b[2], (min(e, b[1]) - max(s, b[0])) / 1e6, b[3]) for b in big if b[0] < e and b[1] > s and b[2] != n]
    print(f"{(s - t0) / 1e6:10.2f} {(e - s) / 1e6:6.2f} q{q} | " + "; ".join(f"{a} {o:.2f} q{qq}" for a, o, qq in ov))
print("# the first encoder convolution (launches over 30 ms), in order: start ms, duration ms, and the summed duration of everything else that overlaps it")
for s, e, n, q, st in ev:
    if "conv3d_split_wino_kernel" in n and e - s > 30_000_000:
        other = sum((min(e, b[1]) - max(s, b[0])) for b in ev if b[0] < e and b[1] > s and b[2] != n) / 1e6
        print(f"{(s - t0) / 1e6:10.2f} {(e - s) / 1e6:7.2f}  overlapped by {other:6.2f} ms of other kernels")
print("# queues seen:", sorted({e[3] for e in ev}), "streams:", sorted({e[4] for e in ev}))
PY
cat $OUT/overlap.txt | cut -c1-230
