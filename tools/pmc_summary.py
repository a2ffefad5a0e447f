#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) into per-kernel HBM traffic.

usage: tools/pmc_summary.py <dir_with_FETCH_SIZE_csv> <dir_with_WRITE_SIZE_csv> <steps_in_run> > profiles/rNN_conv_traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads;
other access patterns must be calibrated on a known byte count: profiles/r01_fetch_calibration.txt (channel_stats / maxpool:
x2.00; conv halo reads = 64-byte pieces at a channel stride: x1.26 for 128-channel inputs, x1.70 for 32-channel inputs)."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


# default 2.0; the split-operand kernels read their halos exactly like the fp32 kernels (64-byte pieces at a channel stride)
# conv3d_split_wino_kernel: calibrated in round 5 (tools/dev/calib_wino_fetch.sh, profiles/r05_ab_experiments.txt section 5): 1.24
# conv3d_split_wino32_kernel (round 6): the x-strip kernel's access pattern (64-byte pieces of a 10 x 10 x 10 halo at a channel stride): its factor, not calibrated separately
FETCH_FACTOR = {"conv3d_split_wino_kernel<true>": 1.24, "conv3d_split_wino32_kernel<true>": 1.42, "conv3d_split_wino32pc_kernel<true>": 1.42, "conv3d_gcr_kernel<2>": 1.26, "conv3d_gcr_kernel<1>": 1.42,
                "conv3d_split_wide_kernel<2, true>": 1.26, "conv3d_split_wide_kernel<2, false>": 1.26,
                "conv3d_split_strip_kernel<true>": 1.42, "conv3d_split_strip_kernel<false>": 1.42,      # the halo reads of conv3d_split_kernel<1>
                "conv3d_split_kernel<1, 2, true, 1>": 1.42, "conv3d_split_kernel<2, 2, true, 1>": 1.42,
                "conv3d_split_kernel<1, 2, false, 1>": 1.42, "conv3d_split_kernel<2, 2, false, 1>": 1.42,
                "conv3d_split_kernel<1, 3, false, 1>": 1.42, "conv3d_split_kernel<2, 3, false, 1>": 1.26}


import glob


def find(d):
    c = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    return c[0] if c else d + "/pmc_counter_collection.csv"


def summarise(fetch_dir, write_dir):
    """-> {"kernels": {name: {launches, fetch_factor, fetch_bytes, write_bytes, hbm_bytes}}} (bytes per launch, averaged over the run)"""
    fetch = load(find(fetch_dir))
    write = load(find(write_dir))
    out = {"units": "bytes per launch (average over the run)", "fetch_correction": "2.0 unless listed per kernel (calibrated)", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        if fetch[k][1] < 1024:
            continue
        n = fetch[k][0]
        fac = FETCH_FACTOR.get(k, 2.0)
        f = fac * fetch[k][1] * 1024 / n
        w = write.get(k, [n, 0.0])[1] * 1024 / n
        out["kernels"][k] = {"launches": n, "fetch_factor": fac, "fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
    return out


if __name__ == "__main__":
    json.dump(summarise(sys.argv[1], sys.argv[2]), sys.stdout, indent=1)
