mkdir -p gpurun_out/r05_chain; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o kt -- python $REPO/bench.py --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate --no-strict-pass --no-host-io-pass > $REPO/gpurun_out/r05_chain/kt_out.json 2> $REPO/gpurun_out/r05_chain/kt_err.txt
cd $REPO
F=$(find /tmp/ktrace -name "*kernel_trace.csv" | head -1)
python - $F <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
prev_end=None
for i,r in enumerate(rows):
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e6 if prev_end else 0
    if gap>5 or 'wino' in r['Kernel_Name']:
        print(f"{(s-t0)/1e6:10.1f} ms gap {gap:8.2f} dur {(e-s)/1e6:8.2f} q{r.get('Queue_Id')} {r['Kernel_Name'][:50]} grid {r.get('Grid_Size_X')}")
    prev_end=max(prev_end or 0,e)
PY
