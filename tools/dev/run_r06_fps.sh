#!/bin/bash
# round 6: the spatially pruned farthest-point kernel: parity tests, A/B timing against the register-resident kernel (GARMENTNETS_FPS_REGIONS=0), and the tail fold's tests
O=gpurun_out/r06_fps; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps or ggm or sa_module or ball_query" 2>&1 | tail -5
echo "--- regions off"; GARMENTNETS_FPS_REGIONS=0 python tools/dev/ab_fps.py 2>&1 | tee $O/ab_off.txt
echo "--- regions on";  GARMENTNETS_FPS_REGIONS=1 python tools/dev/ab_fps.py 2>&1 | tee $O/ab_on.txt
