#!/usr/bin/env python3
"""Condenses tools/collect_batch_pmc.sh's counter files (gpurun_out/<round>_batch_pmc_<size>)
into profiles/<round>_batch_pmc.json: per item size, the wave kernel's mean SQ counters per
launch, instructions per item, and how long the VALU instructions alone take to issue
(a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles; 1024 SIMDs) next to the
kernel's duration under the profiler."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
GO = os.path.join(ROOT, "gpurun_out")
CLOCK_HZ = 2.4e9
SIMDS = 256 * 4
out = {"round": rnd, "clock_hz_assumed": CLOCK_HZ, "simds": SIMDS, "sizes": {}}
for d in sorted(glob.glob(os.path.join(GO, f"{rnd}_batch_pmc_*")), key=lambda p: int(p.rsplit("_", 1)[1]) if p.rsplit("_", 1)[1].isdigit() else -1):
    if not os.path.isdir(d):
        continue
    size = int(d.rsplit("_", 1)[1])
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    dur = {}
    with open(max(files, key=os.path.getmtime)) as f:
        for r in csv.DictReader(f):
            if "crc32_batch_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not agg:
        continue
    mean = {k: sum(v) / len(v) for k, v in agg.items()}
    items = min((1 << 30) // size, 1 << 20)
    kernel_us = sum(dur.values()) / len(dur)
    valu_issue_us = mean["SQ_INSTS_VALU"] * 4 / SIMDS / CLOCK_HZ * 1e6
    out["sizes"][str(size)] = {
        "items_per_launch": items,
        "launches": len(dur),
        "kernel_us_under_profiler": round(kernel_us, 1),
        "counters_per_launch": {k: round(v, 1) for k, v in mean.items()},
        "valu_wave_instr_per_item": round(mean["SQ_INSTS_VALU"] / items, 2),
        "lds_wave_instr_per_item": round(mean["SQ_INSTS_LDS"] / items, 2),
        "valu_issue_us_at_4_cycles_per_instr": round(valu_issue_us, 1),
        "valu_issue_share_of_kernel": round(valu_issue_us / kernel_us, 3),
        "lds_bank_conflict_cycle_share": round(mean["SQ_LDS_BANK_CONFLICT"] / max(mean["SQ_LDS_IDX_ACTIVE"], 1.0), 3),
    }
with open(os.path.join(ROOT, "profiles", f"{rnd}_batch_pmc.json"), "w") as f:
    json.dump(out, f, indent=1)
for k, v in out["sizes"].items():
    print(k, {x: v[x] for x in ("kernel_us_under_profiler", "valu_wave_instr_per_item", "lds_wave_instr_per_item",
                                "valu_issue_us_at_4_cycles_per_instr", "valu_issue_share_of_kernel",
                                "lds_bank_conflict_cycle_share")})
