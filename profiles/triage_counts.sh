#!/bin/bash
# GPU box: dynamic instruction counts of the triage kernels on one input (one counter pass per counter set).
#   bash profiles/triage_counts.sh <tag> <configs[1]|shard3> [lib.so]   ->  gpurun_out/<tag>_triage_counts.txt
set -e
TAG=$1; WHICH=$2; LIB=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export INDELMINER_AMD_LIB=$LIB
O=gpurun_out/tcnt_$TAG; rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O -- python3 profiles/triage_probe.py "$WHICH" > gpurun_out/${TAG}_triage_probe.log 2>&1
python3 profiles/summarize_pmc.py $O triage > gpurun_out/${TAG}_triage_counts.txt
rm -rf $O
cat gpurun_out/${TAG}_triage_counts.txt; tail -2 gpurun_out/${TAG}_triage_probe.log
