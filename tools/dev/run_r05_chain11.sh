mkdir -p gpurun_out/r05_chain; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o kt -- python $REPO/bench.py --no-in-flight-pass --no-latency-b1 --no-pmc --no-cpu-baseline --no-validate > $REPO/gpurun_out/r05_chain/kt_out.json 2> $REPO/gpurun_out/r05_chain/kt_err.txt
cd $REPO
F=$(find /tmp/ktrace -name "*kernel_trace.csv" | head -1)
python - $F <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
w=[r for r in rows if 'wino' in r['Kernel_Name']]
print(len(rows), len(w))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in w]
print('wino durations ms:', ' '.join(f'{x:.1f}' for x in d))
# largest gaps / longest kernels overall
allk=sorted(rows,key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']),reverse=True)[:8]
for r in allk: print(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X'), r.get('Queue_Id'))
PY
python - <<'PY'
import json
for l in open('gpurun_out/r05_chain/kt_out.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['literal_affine']['value'])
PY
