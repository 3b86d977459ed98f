#!/bin/bash
# GPU box: everything the judged numbers come from, for one round tag.
#   bash profiles/collect.sh r01_c   ->  gpurun_out/collect/r01_c_*   (copy into profiles/ afterwards)
# Passes (separately, as the pool requires: --pmc only ever together with --kernel-trace):
#   1. bench.py un-profiled                              -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats                  -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   3-6. rocprofv3 --kernel-trace --pmc <SQ a> / <SQ b> / FETCH_SIZE / WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
#                                                        -> <tag>_pmc_summary.txt, <tag>_traffic.json
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/collect; mkdir -p $O
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/err.log
cp "$(ls $O/stats/*/*kernel_stats.csv | head -1)" $O/${TAG}_kernel_stats.csv; rm -rf $O/stats
: > $O/${TAG}_pmc_summary.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/err_pmc$i.log
  python3 profiles/summarize_pmc.py $O/pmc$i >> $O/${TAG}_pmc_summary.txt
  rm -rf $O/pmc$i
done
python3 - "$O" "$TAG" <<'PY'
import json, re, sys
O, TAG = sys.argv[1:3]
txt = open("%s/%s_pmc_summary.txt" % (O, TAG)).read()
vals = {}
cur = None
for line in txt.splitlines():
    if line and not line.startswith(" "):
        cur = line.strip()
    m = re.match(r"\s+(\w+)\s+mean\s+([\d.]+)", line)
    if m and cur and cur.startswith("realign_kernel"):
        vals[m.group(1)] = float(m.group(2))
b = json.load(open("%s/%s_bench.json" % (O, TAG)))
hbm = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
json.dump({"kernel": "realign_kernel<6, true>",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (another) -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
           "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
           "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B",
           "hbm_bytes_per_launch": hbm,
           "tcc_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
           "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]},
          open("%s/%s_traffic.json" % (O, TAG), "w"), indent=2)
print(open("%s/%s_traffic.json" % (O, TAG)).read())
PY
cut -d, -f1-4 $O/${TAG}_kernel_stats.csv
