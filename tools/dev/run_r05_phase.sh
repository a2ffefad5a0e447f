mkdir -p gpurun_out/r05_chain
for ph in 0 10 20 40 0 20; do echo "== phase $ph"; GARMENTNETS_STRIP_PHASE=$ph timeout 200 python tools/dev/ab_zero.py 2>&1 | grep TF | cut -c1-100; done | tee gpurun_out/r05_chain/strip_phase.txt
