mkdir -p gpurun_out/r05_chain
for i in 1 2; do
GARMENTNETS_BENCH_STALL_TRACE=1 timeout 600 python bench.py --no-in-flight-pass --no-latency-b1 --no-pmc > gpurun_out/r05_chain/st_out$i.json 2> gpurun_out/r05_chain/st_err$i.txt
python - $i <<'PY'
import json,sys
for l in open(f'gpurun_out/r05_chain/st_out{sys.argv[1]}.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
PY
grep "stall-trace" gpurun_out/r05_chain/st_err$i.txt
done
python - <<'PY'
# the distinct main-thread stacks of the headline pass of run 1, with counts
import re,collections
t=open('gpurun_out/r05_chain/st_err1.txt').read().split('[stall-trace]')[0]
blocks=t.split('Timeout (')
c=collections.Counter()
for b in blocks[1:]:
    m=b.split('Thread 0x')
    main=m[-1]
    lines=[l.strip() for l in main.split('\n') if l.strip().startswith('File')][:4]
    c[' | '.join(lines)]+=1
for k,v in c.most_common(12): print(v,k)
PY
