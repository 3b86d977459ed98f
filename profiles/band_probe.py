#!/usr/bin/env python3
"""The banded kernel (-g 1 / 2 / 5 / 12) on the configs[1] candidates, one realign launch per setting (bench.py's other_parameters leg alone).
    [INDELMINER_AMD_LIB=...] python profiles/band_probe.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
r = bench.other_parameters_measure(0)
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join("%s: %.3f ms" % (x["flags"], x["ms"]) for x in r["settings"]), r["identical_to_the_oracle_on_the_samples"], flush=True)
