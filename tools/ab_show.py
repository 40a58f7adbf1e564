"""Side-by-side device times of tools/ab_variants.sh outputs: python tools/ab_show.py prev new wb3 ..."""
import json, re, sys
rows = {}
for v in sys.argv[1:]:
    for i in (1, 2):
        txt = open(f"gpurun_out/abb_{v}{i}.txt").read()
        for m in re.finditer(r"\{[^{}]*\}", txt):
            try:
                r = json.loads(m.group(0))
            except Exception:
                continue
            if "device_us_per_call" not in r:
                continue
            key = (r["kind"], r.get("descriptors", ""), str(r.get("buffer_bytes", r.get("file_bytes", r.get("chunk_data_bytes", "")))), r.get("buffers", r.get("files", "")))
            rows.setdefault(key, {}).setdefault(v, []).append(r["device_us_per_call"])
for key, d in rows.items():
    print("%-22s %-7s %-10s %-8s" % key, "  ".join("%s %7.2f" % (v, min(d[v])) for v in sys.argv[1:] if v in d))
