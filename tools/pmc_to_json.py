#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (tools/pmc_passes.sh) -> one JSON per configuration for profiles/ and bench.py.

usage: pmc_to_json.py <dir with pass*/p_counter_collection.csv> <out.json> <example> <N> <H>
Per-ITERATION totals: a reverse_once launches every rollout-kernel instantiation it uses once (the split launch of H1 /
Allegro: the even launch + the one-wavefront launch), so the counters are averaged per dispatch and kernel NAME and then
summed over the names.  FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB = 1024 B by rocprofv3 on this
stack (calibrated in round 1 against the known byte counts of the launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out, example, N, H = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "rollout_kernel" not in row["Kernel_Name"]:
            continue
        a = acc[row["Counter_Name"]][row["Kernel_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
tot = {c: sum(v[0] / v[1] for v in names.values()) for c, names in acc.items()}
kernels = sorted({k for names in acc.values() for k in names})
wave_steps = (N + 1) * (H + 1)
d = {"example": example, "Nsample": N, "Hsample": H, "kernels": [k[:160] for k in kernels],
     "source": f"rocprofv3 --pmc passes of `bench.py --example {example}` (tools/pmc_passes.sh), per-iteration totals",
     "counters": tot}
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    d["FETCH_SIZE_KB"], d["WRITE_SIZE_KB"] = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
    d["hbm_bytes_per_launch"] = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
if "SQ_INSTS_VALU" in tot:
    d["valu_insts_per_wave_env_step"] = tot["SQ_INSTS_VALU"] / wave_steps
    d["salu_insts_per_wave_env_step"] = tot.get("SQ_INSTS_SALU", 0) / wave_steps
    d["lds_insts_per_wave_env_step"] = tot.get("SQ_INSTS_LDS", 0) / wave_steps
    d["branch_insts_per_wave_env_step"] = tot.get("SQ_INSTS_BRANCH", 0) / wave_steps
    classes = ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "CVT")
    if all("SQ_INSTS_VALU_" + k in tot for k in classes):
        mix = {k: tot["SQ_INSTS_VALU_" + k] / tot["SQ_INSTS_VALU"] for k in classes}
        mix["other (mov, cndmask, cmp, readlane, dpp, bit ops)"] = 1.0 - sum(mix.values())
        d["valu_mix"] = mix
        d["valu_full_rate_fp32_frac"] = mix["FMA_F32"] + mix["ADD_F32"] + mix["MUL_F32"]
    else:   # the per-class pass (tools/pmc_passes.sh: pass 6) was not run: null, never a 0.0 that reads as data
        d["valu_mix"] = None
        d["valu_full_rate_fp32_frac"] = None
if "SQ_THREAD_CYCLES_VALU" not in tot:
    d["valu_active_lanes_per_inst"] = None
    d["valu_lane_utilisation"] = None
if "SQ_THREAD_CYCLES_VALU" in tot and tot.get("SQ_INSTS_VALU"):
    d["valu_active_lanes_per_inst"] = tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_INSTS_VALU"]
    d["valu_lane_utilisation"] = d["valu_active_lanes_per_inst"] / 64.0
if "SQ_WAVE_CYCLES" in tot:
    wc = tot["SQ_WAVE_CYCLES"]
    d["wave_time_breakdown"] = {"issuing (SQ_ACTIVE_INST_ANY)": tot.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                "parked at s_waitcnt (SQ_WAIT_ANY)": tot.get("SQ_WAIT_ANY", 0) / wc,
                                "issue stall (SQ_WAIT_INST_ANY)": tot.get("SQ_WAIT_INST_ANY", 0) / wc,
                                "of which LDS issue stall (SQ_WAIT_INST_LDS)": tot.get("SQ_WAIT_INST_LDS", 0) / wc}
    d["note_units"] = "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); ratios are unit-free"
json.dump(d, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k not in ("counters", "kernels")}, indent=1))
