#!/bin/bash
# round 6 (re-entry): the whole GPU suite + the driver's bench invocation on one box
O=gpurun_out/r06_sanity; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/gputest.log 2>&1; tail -4 $O/gputest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default.json 2> $O/default.err; echo "default rc $?"; cp gpurun_out/bench_detail.json $O/default_detail.json
tail -c 2600 $O/default.json
