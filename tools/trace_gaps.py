#!/usr/bin/env python3
"""tools/trace_gaps.py — per-kernel time and inter-kernel gaps from a rocprofv3 --kernel-trace csv
(last N dispatches of the gcn:: kernels), to see where a step's wall time goes."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "gcn::" in r["Kernel_Name"]]
tail = ours[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void gcn::", "")[:34]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {name}")
    busy += e - s
    prev_end = e
span = int(tail[-1]["End_Timestamp"]) - t0
print(f"span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({100.0 * busy / span:.1f} %)")
