#!/bin/bash
# GPU box: dynamic instruction counts of the realign kernel on the configs[1] candidates (12 232 reads), one counter pass.
#   bash profiles/realign_counts.sh <tag> [lib.so]   ->  gpurun_out/<tag>_counts.txt
set -e
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export INDELMINER_AMD_LIB=$LIB
O=gpurun_out/cnt_$TAG; rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O -- python3 profiles/scaling_probe.py 12232 > gpurun_out/${TAG}_probe.log 2>&1
python3 profiles/summarize_pmc.py $O realign > gpurun_out/${TAG}_counts.txt
rm -rf $O
python3 - <<PY
import re
v={}
for l in open("gpurun_out/${TAG}_counts.txt"):
    m=re.match(r"\s+(\w+)\s+mean\s+([\d.]+)",l)
    if m: v[m.group(1)]=float(m.group(2))
n=12232.0
print("${TAG}: per read  VALU %.1f  SALU %.1f  LDS %.1f  BRANCH %.1f  wave-cycles %.0f  wait %.0f%%" % (v["SQ_INSTS_VALU"]/n, v["SQ_INSTS_SALU"]/n, v["SQ_INSTS_LDS"]/n, v["SQ_INSTS_BRANCH"]/n, v["SQ_WAVE_CYCLES"]/n, 100*v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"]))
PY
cat gpurun_out/${TAG}_probe.log | tail -2
