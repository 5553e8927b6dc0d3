#!/usr/bin/env python3
"""tools/pmc_table.py <dir> <kernel-substring>: per-kernel averages of every counter found under <dir>/pmc_*/."""
import collections
import csv
import glob
import sys

root, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{root}/pmc_*/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    d = dur.get(k, [0])
    print(f"== {k}: avg duration under PMC {sum(d) / len(d):.1f} us over {len(d)} launches")
    for c in sorted(v):
        print(f"   {c:42s} {sum(v[c]) / len(v[c]):18.1f}   (n={len(v[c])})")
