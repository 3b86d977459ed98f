#!/bin/bash
# GPU box: per-kernel times of the bench step.  usage: bash profiles/run_stats.sh <tag>  -> gpurun_out/prof/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof/$1_bench_under_rocprof.json 2> gpurun_out/prof/err.log || exit 1
f=$(ls gpurun_out/prof/stats/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/prof/$1_kernel_stats.csv; rm -rf gpurun_out/prof/stats
cut -d, -f1-4,6,7 gpurun_out/prof/$1_kernel_stats.csv
