#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_diag.sh: pmc_summary.py OUTDIR"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("digit_pass_kernel", "global_histogram_banked", "global_histogram_scan", "morton_encode30", "bucket_histogram"):
        if key in name:
            return key
    return None


def main():
    out = sys.argv[1]
    table = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", ""))
                if k:
                    table[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in table.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
