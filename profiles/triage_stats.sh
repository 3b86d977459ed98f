#!/bin/bash
# usage: stats_probe.sh tag which [lib]
TAG=$1; WHICH=$2; LIB=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export INDELMINER_AMD_LIB=$LIB
O=gpurun_out/tst_$TAG; rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 profiles/triage_probe.py "$WHICH" > gpurun_out/${TAG}_probe.log 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1); echo "== $TAG"; cut -d, -f1-4,6,7 $f | grep -i "triage\|Name"; tail -1 gpurun_out/${TAG}_probe.log; rm -rf $O
