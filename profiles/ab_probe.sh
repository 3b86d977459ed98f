#!/bin/bash
# GPU box: the realign kernel's duration for several builds of the library on the SAME box, interleaved twice
# (box-to-box spread is +-5 %, more than most single changes).   bash profiles/ab_probe.sh tag lib1.so lib2.so ...
TAG=$1; shift
O=gpurun_out/${TAG}_ab.log; : > $O
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)" >> $O
    INDELMINER_AMD_LIB=$L python3 profiles/scaling_probe.py 12232 73153 >> $O 2>&1
  done
done
cat $O
