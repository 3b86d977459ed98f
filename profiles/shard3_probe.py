#!/usr/bin/env python3
"""bench.py's config-3 shard measurement on its own (for rocprofv3 --kernel-trace --stats / --pmc passes):
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02_shard3 -- python3 profiles/shard3_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.shard3_measure(0, steps=int(os.environ.get("STEPS", "24")))))
