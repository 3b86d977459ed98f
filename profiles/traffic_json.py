#!/usr/bin/env python3
"""HBM bytes per realign launch from a summarize_pmc.py summary (FETCH_SIZE / WRITE_SIZE passes), with the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B: x2), beside the algorithmic bytes of
the same launch taken from the bench (or shard probe) JSON line.
  traffic_json.py <pmc_summary.txt> <bench.json> <out.json> "<profiled command>" """
import json
import re
import sys

summary, bench_json, out, cmd = sys.argv[1:5]
vals = {}
tri = {}
cur = None
for line in open(summary).read().splitlines():
    if line and not line.startswith(" "):
        cur = line.strip()
    m = re.match(r"\s+(\w+)\s+mean\s+([\d.]+)", line)
    if m and cur and cur.startswith("realign_kernel"):
        vals[m.group(1)] = float(m.group(2))
    if m and cur and cur.startswith("triage_classify_kernel"):
        tri[m.group(1)] = float(m.group(2))
b = json.loads(open(bench_json).read().strip().splitlines()[-1])
alg = b["roofline"]["algorithmic_bytes_per_launch"] if "roofline" in b else b["realign"]["algorithmic_bytes_per_launch"]
hbm = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
doc = {"kernel": "realign_kernel<6, true>",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (another) -- " + cmd,
       "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
       "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B",
       "hbm_bytes_per_launch": hbm,
       "tcc_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
       "algorithmic_bytes_per_launch": alg}
if "FETCH_SIZE" in tri:
    doc["triage_classify_kernel"] = {"FETCH_SIZE_KB_per_launch": tri["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": tri.get("WRITE_SIZE"),
                                     "hbm_bytes_per_launch": int(tri["FETCH_SIZE"] * 1024 * 2 + tri.get("WRITE_SIZE", 0) * 1024)}
json.dump(doc, open(out, "w"), indent=2)
print(open(out).read())
