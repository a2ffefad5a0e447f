#!/bin/bash
# dev-only (GPU box): FETCH_SIZE calibration of the Winograd conv kernel -> gpurun_out/r05_ab/wino_fetch_calibration.txt
REPO=$(cd "$(dirname "$0")/../.." && pwd); O=$REPO/gpurun_out/r05_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/calib_wino
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/calib_wino -o pmc -- python $REPO/tools/dev/calib_wino_fetch.py > $O/calib_wino.log 2>&1 < /dev/null
python - <<'PY' > $O/wino_fetch_calibration.txt
import csv, glob
p = glob.glob("/tmp/calib_wino/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if "wino" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
known = 8192 * 256 * 128 * 4 / 1024
print("conv3d_split_wino_kernel<true>, Cin=128: 16 B/lane, 64-B pieces at a 512-B stride + 1-KB weight-fragment DMAs (L2-resident pack)")
for x in v: print(f"  FETCH_SIZE {x:.0f} KB   known input {known:.0f} KB   factor {known / x:.3f}")
PY
cat $O/wino_fetch_calibration.txt
