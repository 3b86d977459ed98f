#!/bin/bash
# Dynamic instruction count per phase of realign_kernel<6,true>.
#   build (here):   bash profiles/phase_counts.sh build     -> exp_libs/stop_<n>.so, n = 0..14, and exp_libs/stop_full.so
#   run (GPU box):  bash profiles/phase_counts.sh run       -> gpurun_out/phase/counts.txt
# Every read of build n stops after phase n (IM_STOP_AFTER, im_realign.hip), so SQ_INSTS_* of build n minus
# build n-1 is what phase n executes.  Diagnostic only; the product library has no such switch.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
PHASES="0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 full"
if [ "$1" = build ]; then
  mkdir -p exp_libs/obj
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iindelminer_amd/csrc"
  for s in im_realign_long im_realign_any im_results im_cluster im_depth im_support im_triage im_flush im_flushwide im_capi im_comm; do hipcc $FL -c indelminer_amd/csrc/$s.hip -o exp_libs/obj/$s.o & done; wait
  build_one() { n=$1; d=""; [ $n != full ] && d="-DIM_STOP_AFTER=$n"; hipcc $FL $d -c indelminer_amd/csrc/im_realign.hip -o exp_libs/obj/realign_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC exp_libs/obj/realign_$n.o exp_libs/obj/im_realign_long.o exp_libs/obj/im_realign_any.o exp_libs/obj/im_results.o exp_libs/obj/im_cluster.o exp_libs/obj/im_depth.o exp_libs/obj/im_support.o exp_libs/obj/im_triage.o exp_libs/obj/im_flush.o exp_libs/obj/im_flushwide.o exp_libs/obj/im_capi.o exp_libs/obj/im_comm.o -ldl -o exp_libs/stop_$n.so; }
  i=0; for n in $PHASES; do build_one $n & i=$((i+1)); [ $((i % 6)) = 0 ] && wait; done; wait
  ls exp_libs/stop_*.so | wc -l
else
  cd /tmp && export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out/phase; : > gpurun_out/phase/counts.txt
  for n in $PHASES; do
    export INDELMINER_AMD_LIB=exp_libs/stop_$n.so
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/phase/p_$n -- python3 profiles/scaling_probe.py 48184 > gpurun_out/phase/p_$n.log 2>&1 || exit 1
    echo "== stop after $n" >> gpurun_out/phase/counts.txt
    python3 profiles/summarize_pmc.py gpurun_out/phase/p_$n realign >> gpurun_out/phase/counts.txt
    rm -rf gpurun_out/phase/p_$n
  done
  cat gpurun_out/phase/counts.txt
fi
