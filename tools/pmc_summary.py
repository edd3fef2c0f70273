#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output (…_counter_collection.csv) per kernel and counter.

usage: pmc_summary.py <dir-or-csv> [kernel-substring]
Prints, per kernel, the number of dispatches and the per-dispatch mean of every counter found.
"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [root] if os.path.isfile(root) else glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float)); disp = defaultdict(set)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if filt and filt not in k: continue
                k = k.split("(")[0][:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add((f, row.get("Dispatch_Id")))
    for k in sorted(acc):
        n = max(1, len(disp[k]))
        print(f"{k}  dispatches={n}")
        for c in sorted(acc[k]):
            print(f"    {c:32s} {acc[k][c] / n:18.1f} per dispatch")


if __name__ == "__main__":
    main()
