#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/profile_round.sh into per-kernel matrix-core utilisation and LDS pressure.

usage: tools/pmc_mfma_summary.py gpurun_out/prof_<tag> > profiles/<tag>_mfma_util.json
Per kernel (sums over its launches in the run):
  mfma_util          = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (BUSY_CYCLES counts cycles per SIMD, 32 per
                       v_mfma_*_32x32x16: MI355X_MICROARCH.md per-instruction table; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8
                       XCDs -- checked against the known MFMA count of the conv: 5.84e8 instructions x 32 cycles per launch)
  effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration (profiled passes clock lower than un-profiled runs)
  mfma_ops_per_wave_cycle, lds instructions per MFMA op, LDS bank-conflict cycles / LDS instructions, LDS-issue stall share of wave cycles
SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (same table)."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (name, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key)
                launches[name] += 1
                try:
                    agg[name]["_duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    return agg, launches


root = sys.argv[1]
mf, n1 = load(os.path.join(root, "mfma"))
iss, n2 = load(os.path.join(root, "issue"))
out = {"source": "rocprofv3 --kernel-trace --pmc (two SQ passes) of bench.py's default workload, tools/profile_round.sh",
       "notes": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); profiled passes clock ~5 % lower than un-profiled runs",
       "kernels": {}}
for k in sorted(mf, key=lambda k: -mf[k].get("GRBM_GUI_ACTIVE", 0.0)):
    c = mf[k]
    act = c.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
        continue
    busy, mops, lds, conf = c["SQ_VALU_MFMA_BUSY_CYCLES"], c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0), c.get("SQ_INSTS_LDS", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    dur = c.get("_duration_ns", 0.0)
    e = {"launches": n1[k], "gui_active_cycles_sum_over_xcds": act, "mfma_busy_cycles": busy, "mfma_util": busy / (1024.0 * act / 8.0),
         "effective_clock_ghz": (act / 8.0) / dur if dur > 0 else None, "duration_ms_per_launch": dur / 1e6 / n1[k] if dur > 0 else None,
         "mfma_mops_f16": mops, "lds_insts": lds, "lds_insts_per_512_mfma_mops": (lds / (mops / 512.0)) if mops else None,
         "lds_bank_conflict_cycles_per_lds_inst": conf / lds if lds else None,
         "lds_issue_stall_share_of_wave_cycles": c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
    i = iss.get(k)
    if i and i.get("SQ_WAIT_ANY") is not None:
        tot = i.get("SQ_ACTIVE_INST_ANY", 0.0) + i.get("SQ_WAIT_ANY", 0.0) + i.get("SQ_WAIT_INST_ANY", 0.0)
        if tot > 0:
            e.update(wave_cycles_issuing=i.get("SQ_ACTIVE_INST_ANY", 0.0) / tot, wave_cycles_parked_waitcnt_or_barrier=i.get("SQ_WAIT_ANY", 0.0) / tot,
                     wave_cycles_issue_stalled=i.get("SQ_WAIT_INST_ANY", 0.0) / tot, valu_insts=i.get("SQ_INSTS_VALU", 0.0), waves=i.get("SQ_WAVES", 0.0))
    out["kernels"][k] = e
json.dump(out, sys.stdout, indent=1)
