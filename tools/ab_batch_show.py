"""Side-by-side device times of the batch_bench rows written by tools/ab_batch.sh."""
import glob
import json
import sys

runs = {}
for f in sorted(glob.glob("gpurun_out/abb_*.txt")):
    tag = f.split("abb_")[1].split(".")[0]
    runs[tag] = json.load(open(f))["rows"]
tags = sorted(runs)
n = len(runs[tags[0]])
for i in range(n):
    r0 = runs[tags[0]][i]
    key = r0["kind"] + " " + str(r0.get("buffer_bytes", r0.get("chunk_data_bytes", r0.get("file_bytes")))) + " " + r0.get("descriptors", "")
    vals = []
    for t in tags:
        r = runs[t][i]
        vals.append(r.get("device_us_per_call"))
    print(f"{key:45s} " + "  ".join(f"{t}={v}" for t, v in zip(tags, vals)))
