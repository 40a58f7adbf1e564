#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into the small, committed
summaries under profiles/ (per round), and writes profiles/traffic.json, which
bench.py reports as roofline.traffic.

  python tools/summarize_profiles.py r01
expects gpurun_out/<round>_trace, _pmc_fetch, _pmc_write, _pmc_sq (see
profiles/README.md for the commands that produce them).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
GO = os.path.join(ROOT, "gpurun_out")
PR = os.path.join(ROOT, "profiles")
os.makedirs(PR, exist_ok=True)
KERNEL = "crc32_tiles_kernel"


def one(pattern):
    hits = glob.glob(os.path.join(GO, pattern))
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return max(hits, key=os.path.getmtime)  # gpurun merges runs into one tree: take the newest


# kernel-trace --stats summary: keep our kernels verbatim, shorten torch's names
with open(one(f"{rnd}_trace/*/*kernel_stats.csv")) as f:
    rows = list(csv.DictReader(f))
with open(os.path.join(PR, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]
        w.writerow(r)
stats = next(r for r in rows if KERNEL in r["Name"])

pmc = {}
for tag in ("fetch", "write", "sq"):
    with open(one(f"{rnd}_pmc_{tag}/*/*counter_collection.csv")) as f:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}

bench = {}
bj = os.path.join(GO, f"bench_{rnd}.json")
if os.path.exists(bj):
    with open(bj) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])

alg = bench.get("roofline", {}).get("algorithmic_bytes_per_launch", 1 << 30)
# MI355X_MICROARCH.md "HBM": FETCH_SIZE is in KiB and, on gfx950, reports exactly
# half of the bytes of a wide (16 B/lane) coalesced streaming read -> double it.
fetch_bytes = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
write_bytes = pmc["WRITE_SIZE"]["mean_per_launch"] * 1024
traffic = {
    "round": rnd,
    "workload": bench.get("config", {}).get("workload"),
    "kernel": stats["Name"],
    "FETCH_SIZE_KiB_raw": pmc["FETCH_SIZE"]["mean_per_launch"],
    "WRITE_SIZE_KiB_raw": pmc["WRITE_SIZE"]["mean_per_launch"],
    "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
    "hbm_bytes_per_launch": fetch_bytes + write_bytes,
    "algorithmic_bytes_per_launch": alg,
    "ratio_to_algorithmic": (fetch_bytes + write_bytes) / alg,
}
with open(os.path.join(PR, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)

# per-launch durations of the tile kernel from the same trace: the --stats average
# takes in the rare launch that is descheduled for milliseconds, the median does not
durs = []
with open(one(f"{rnd}_trace/*/*kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        if KERNEL in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
durs.sort()
trimmed = durs[: max(1, len(durs) * 99 // 100)]  # without the slowest 1 %

# the same command as the committed bench line (--steps 20 --warmup 5) under the profiler:
# its LAST 20 launches are the timed steps, the ones bench.py's own events time
steps20 = None
hits20 = glob.glob(os.path.join(GO, f"{rnd}_trace20/*/*kernel_trace.csv"))
if hits20:
    rows20 = []
    first_other = None  # the secondary rows start with the first batch / piece kernel launch
    with open(hits20[0]) as f:
        allrows = list(csv.DictReader(f))
    for r in allrows:
        if ("crc32_batch_kernel" in r["Kernel_Name"] or "crc32_pieces_kernel" in r["Kernel_Name"]):
            t = int(r["Start_Timestamp"])
            first_other = t if first_other is None else min(first_other, t)
    for r in allrows:
        if KERNEL in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            t0, d = int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if d > 100.0 and (first_other is None or t0 < first_other):  # a whole-GiB launch of the main phase
                rows20.append((t0, d))
    rows20.sort()
    # the main phase's launches end with the 64 untimed launches of the 64 MiB / 1 MiB rows' set-up? no: those
    # are shorter than 100 us; its last 20 whole-GiB launches are the timed steps
    last = sorted(d for _, d in rows20[-20:])
    if last:
        steps20 = {"launches_in_trace": len(rows20), "timed_launches": len(last), "avg_us": sum(last) / len(last),
                   "median_us": last[len(last) // 2], "min_us": last[0], "max_us": last[-1]}

vec = alg / 1024.0  # wave-level 1 KiB loads per launch
summary = {
    "round": rnd,
    "bench_line": bench,
    "rocprof_kernel_trace": {
        "kernel": stats["Name"], "calls": int(stats["Calls"]),
        "avg_us": float(stats["AverageNs"]) / 1e3, "min_us": float(stats["MinNs"]) / 1e3,
        "max_us": float(stats["MaxNs"]) / 1e3,
        "GBps_at_avg": alg / float(stats["AverageNs"]),
        "median_us": durs[len(durs) // 2] if durs else None,
        "p90_us": durs[len(durs) * 9 // 10] if durs else None,
        "avg_us_without_slowest_1pct": sum(trimmed) / len(trimmed) if durs else None,
        "launches_over_1ms": sum(1 for d in durs if d > 1000.0),
    },
    "rocprof_kernel_trace_steps20_timed_launches": steps20,
    "pmc_per_launch": pmc,
    "derived": {
        "valu_wave_instr_per_1KiB_wave_load": pmc["SQ_INSTS_VALU"]["mean_per_launch"] / vec,
        "lds_wave_instr_per_1KiB_wave_load": pmc["SQ_INSTS_LDS"]["mean_per_launch"] / vec,
        "lds_bank_conflict_cycle_share": pmc["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"]
        / pmc["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"],
        "wave_cycles_waiting_share": pmc["SQ_WAIT_ANY"]["mean_per_launch"] / pmc["SQ_WAVE_CYCLES"]["mean_per_launch"],
        "wave_cycles_issue_stall_share": pmc["SQ_WAIT_INST_ANY"]["mean_per_launch"]
        / pmc["SQ_WAVE_CYCLES"]["mean_per_launch"],
    },
    "traffic": traffic,
}
with open(os.path.join(PR, f"{rnd}_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary["rocprof_kernel_trace"], indent=1))
print(json.dumps(steps20, indent=1))
print(json.dumps(summary["derived"], indent=1))
print(json.dumps(traffic, indent=1))

# Every kernel of the traced bench command (bench.py --steps 20 --warmup 5, secondary rows
# included), as clusters of like durations per kernel: each secondary row of the bench line
# times one kernel on one workload 20 times (plus its warm-up calls), so its kernel_us must
# be the median of one cluster here.
if hits20:
    by = collections.defaultdict(list)
    with open(hits20[0]) as f:
        for r in csv.DictReader(f):
            if "crc32_" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void sift::", "").replace("sift::", "")
                by[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    clusters = {}
    for name, ds in by.items():
        ds.sort()
        groups, cur = [], [ds[0]]
        for d in ds[1:]:
            if d > cur[0] * 1.25:  # a new cluster starts 25 % above the last one's smallest member
                groups.append(cur)
                cur = [d]
            else:
                cur.append(d)
        groups.append(cur)
        clusters[name] = [{"launches": len(g), "median_us": round(g[len(g) // 2], 2), "min_us": round(g[0], 2),
                           "max_us": round(g[-1], 2)} for g in groups if len(g) >= 5]
    with open(os.path.join(PR, f"{rnd}_bench_kernel_clusters.json"), "w") as f:
        json.dump({"round": rnd, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 "
                                            "--no-cpu-baseline", "kernels": clusters}, f, indent=1)
    hs = glob.glob(os.path.join(GO, f"{rnd}_trace20/*/*kernel_stats.csv"))
    if hs:
        with open(max(hs, key=os.path.getmtime)) as f:
            srows = [r for r in csv.DictReader(f) if "crc32_" in r["Name"]]
        with open(os.path.join(PR, f"{rnd}_bench_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=srows[0].keys())
            w.writeheader()
            for r in srows:
                r["Name"] = r["Name"][:120]
                w.writerow(r)

# the other files tools/collect_profiles.sh leaves under gpurun_out/
import shutil

for src, dst in ((f"bench_{rnd}.json", f"{rnd}_bench.json"), (f"{rnd}_batch_bench.json", f"{rnd}_batch_bench.json"),
                 (f"{rnd}_size_sweep.json", f"{rnd}_size_sweep.json"), (f"{rnd}_async_sweep.json", f"{rnd}_async_sweep.json"),
                 (f"{rnd}_group_trace.txt", f"{rnd}_group_trace.txt"), (f"{rnd}_cut_grid.json", f"{rnd}_cut_grid.json"),
                 (f"{rnd}_lone_long_item.txt", f"{rnd}_lone_long_item.txt")):
    if os.path.exists(os.path.join(GO, src)):
        shutil.copyfile(os.path.join(GO, src), os.path.join(PR, dst))
hits = glob.glob(os.path.join(GO, f"{rnd}_batch_trace/*/*kernel_stats.csv"))
if hits:
    with open(max(hits, key=os.path.getmtime)) as f:
        brows = list(csv.DictReader(f))
    with open(os.path.join(PR, f"{rnd}_batch_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=brows[0].keys())
        w.writeheader()
        for r in brows:
            r["Name"] = r["Name"][:120]
            w.writerow(r)
