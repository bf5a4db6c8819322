#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: per kernel, mean counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            a = acc[k][c]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        # rocprofv3 emits one row per (dispatch, counter[, dimension]); report the per-dispatch mean
        print("    %-24s mean/dispatch %16.1f   (rows %d)" % (c, s / max(n, 1), n))
