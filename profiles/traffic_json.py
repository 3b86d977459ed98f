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
# what binds the kernels: instruction issue (SQ counter passes of the same summary).  GRBM_GUI_ACTIVE counts per XCD (8 of them);
# 1024 SIMDs; a wave-wide VALU instruction holds its SIMD's issue port for 4 cycles.
def issue(v, units):
    if "SQ_INSTS_VALU" not in v or not units:
        return None
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    d = {"valu_per_read": v["SQ_INSTS_VALU"] / units, "salu_per_read": v["SQ_INSTS_SALU"] / units, "lds_per_read": v["SQ_INSTS_LDS"] / units,
         "valu_busy_frac": (v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cyc)) if cyc else None,
         "valu_active_over_wave_cycles": v["SQ_ACTIVE_INST_VALU"] * 4.0 / v["SQ_WAVE_CYCLES"],
         "wait_frac_of_wave_cycles": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
         "lds_conflict_frac": (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]) if v.get("SQ_LDS_IDX_ACTIVE") else None,
         "kernel_cycles": cyc, "units_per_launch": units}
    return d
cfg = b.get("config", b)
n_cand = cfg.get("candidates_per_step")
n_rec = cfg.get("reads_per_step")
doc["issue"] = issue(vals, n_cand)
if doc["issue"]:
    doc["issue"]["unit"] = "candidate read"
    doc["issue"]["source"] = "rocprofv3 --kernel-trace --pmc <SQ counters> (two passes) -- " + cmd
ti = issue(tri, n_rec)
if ti and "triage_classify_kernel" in doc:
    ti["unit"] = "delivered record"
    doc["triage_classify_kernel"]["issue"] = ti
json.dump(doc, open(out, "w"), indent=2)
print(open(out).read())
