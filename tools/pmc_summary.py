#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
            if "anonymous" in row["Kernel_Name"]:
                name = row["Kernel_Name"].split("::")[1].split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", os.path.relpath(path, root))
    for k, counters in acc.items():
        if not k.startswith("k_"):
            continue
        parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(counters.items())]
        n = len(next(iter(counters.values())))
        print(f"  {k} (n={n}): " + " ".join(parts))
