#!/bin/bash
# round 6: timing-only ablations of the 32-wide Winograd kernel, one box
echo "== shipped"; python tools/dev/ab_wino32.py w32 2>&1 | grep "B="
for v in "$@"; do echo "== $v"; GARMENTNETS_HIP_LIB=tools/dev/_build/lib_w32_$v.so python tools/dev/ab_wino32.py w32 2>&1 | grep "B="; done
