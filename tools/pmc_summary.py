#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel and counter, launches and the average counter value per launch.
   python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel substring ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2:]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "").split("(")[0]
            k = k.replace("kzg::", "").replace("void ", "").strip()
            if want and not any(w in k for w in want):
                continue
            c = row["Counter_Name"]
            a = acc[k][c]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {k: {c: {"launches": v[0], "avg": v[1] / v[0]} for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
