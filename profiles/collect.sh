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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-shard3 > $O/${TAG}_bench_under_rocprof.json 2> $O/err.log
cp "$(ls $O/stats/*/*kernel_stats.csv | head -1)" $O/${TAG}_kernel_stats.csv; rm -rf $O/stats
: > $O/${TAG}_pmc_summary.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-shard3 > /dev/null 2> $O/err_pmc$i.log
  python3 profiles/summarize_pmc.py $O/pmc$i >> $O/${TAG}_pmc_summary.txt
  rm -rf $O/pmc$i
done
python3 profiles/traffic_json.py $O/${TAG}_pmc_summary.txt $O/${TAG}_bench.json $O/${TAG}_traffic.json "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-shard3"
cut -d, -f1-4 $O/${TAG}_kernel_stats.csv
# the config-3 per-GPU shard (bench.py's shard_config3 object) on its own: kernel stats, then the same counter passes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 profiles/shard3_probe.py > $O/${TAG}_shard3_under_rocprof.json 2> $O/err3.log
cp "$(ls $O/stats3/*/*kernel_stats.csv | head -1)" $O/${TAG}_shard3_kernel_stats.csv; rm -rf $O/stats3
: > $O/${TAG}_shard3_pmc_summary.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  STEPS=8 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc3_$i -- python3 profiles/shard3_probe.py > /dev/null 2> $O/err_pmc3_$i.log
  python3 profiles/summarize_pmc.py $O/pmc3_$i >> $O/${TAG}_shard3_pmc_summary.txt
  rm -rf $O/pmc3_$i
done
python3 profiles/traffic_json.py $O/${TAG}_shard3_pmc_summary.txt $O/${TAG}_shard3_under_rocprof.json $O/${TAG}_shard3_traffic.json "STEPS=8 python3 profiles/shard3_probe.py"
cut -d, -f1-4 $O/${TAG}_shard3_kernel_stats.csv
