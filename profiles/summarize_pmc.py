#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

def main(root, kernel_filter=None):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("im::(anonymous namespace)::", "").split("(")[0]
            if kernel_filter and kernel_filter not in name:
                continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-24s mean %16.1f  n=%d" % (c, sum(v) / len(v), len(v)))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
