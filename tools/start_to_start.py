#!/usr/bin/env python3
"""Kernel duration and start-to-start interval of back-to-back launches, from rocprofv3
--kernel-trace CSVs (one directory per size, written by tools/collect_midsize.sh):
  python tools/start_to_start.py <round>   ->  profiles/<round>_midsize_attribution.json
For each size: the tile kernel's dispatches in time order; duration = End - Start of a
dispatch, interval = Start of the next minus Start of this one (medians over the
steady-state part of the run).  interval - duration is the time no kernel of the
stream is running: dispatch overhead of the command processor when the host keeps
up, host issue time when it does not (compare with issue_us_per_call from the same run)."""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
GO = os.path.join(ROOT, "gpurun_out")
rows = []
for d in sorted(glob.glob(os.path.join(GO, f"{rnd}_mid_*")), key=lambda p: int(p.rsplit("_", 1)[1]) if p.rsplit("_", 1)[1].isdigit() else -1):
    if not os.path.isdir(d):
        continue
    size = int(d.rsplit("_", 1)[1])
    files = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    if not files:
        continue
    ev = []
    with open(max(files, key=os.path.getmtime)) as f:
        for r in csv.DictReader(f):
            if "crc32_tiles_kernel" in r["Kernel_Name"]:
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ev.sort()
    # steady state: intervals under 10x the median (drops the pauses between repetitions)
    iv = [(b[0] - a[0], a[1] - a[0], b[0] - a[1]) for a, b in zip(ev, ev[1:])]
    med = statistics.median(x[0] for x in iv)
    iv = [x for x in iv if x[0] < 10 * med]
    row = {"bytes": size, "launches": len(ev),
           "kernel_duration_us_median": round(statistics.median(x[1] for x in iv) / 1e3, 2),
           "start_to_start_us_median": round(statistics.median(x[0] for x in iv) / 1e3, 2),
           "gap_end_to_next_start_us_median": round(statistics.median(x[2] for x in iv) / 1e3, 2)}
    jf = os.path.join(GO, f"{rnd}_mid_{size}.json")
    if os.path.exists(jf):
        try:
            with open(jf) as f:
                r0 = json.load(f)["rows"][0]
            row["host_issue_us_per_call_under_profiler"] = r0["issue_us_per_call"]
            row["wall_us_per_call_under_profiler"] = r0["wall_us_per_call"]
        except (ValueError, KeyError, IndexError):
            pass
    rows.append(row)
plain = os.path.join(GO, f"{rnd}_mid_plain.json")
out = {"round": rnd, "rows": rows}
if os.path.exists(plain):
    with open(plain) as f:
        out["without_profiler"] = json.load(f)["rows"]
with open(os.path.join(ROOT, "profiles", f"{rnd}_midsize_attribution.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
