#!/usr/bin/env python3
"""tools/pmc_summary.py — fold rocprofv3 --pmc csv outputs into a small JSON under profiles/.

Collection (on the GPU box, one pass per counter group — TCC has 4 slots, FETCH_SIZE costs 3,
WRITE_SIZE 2; never combined with the hip/hsa trace domains):

    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_<tag>_$c -- \
          python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    done
    python tools/pmc_summary.py --glob 'gpurun_out/pmc_<tag>_*' --graph reddit --k 128 \
        --launches-per-spmm 2 --out profiles/pmc_latest.json

Corrections follow MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) = TCC_EA0_RDREQ x 64 B reports exactly
half of the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE is exact.  The bytes
are L2-miss (fabric-side) traffic: Infinity-Cache hits are included.
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glob", required=True)
    ap.add_argument("--graph", default="reddit")
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--launches-per-spmm", type=int, default=1)
    ap.add_argument("--algorithmic-bytes-per-launch", type=int, default=0)
    ap.add_argument("--kernel", default="spmm_chunk_kernel")
    ap.add_argument("--order", default="none")
    ap.add_argument("--source", default="", help="where the raw counter tables of this summary are kept (profiles/...)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--merge-into", default="", help="profiles/pmc_latest.json: replace the entry of this (graph, k, order)")
    args = ap.parse_args()

    agg = collections.defaultdict(list)
    names = set()
    for d in sorted(glob.glob(args.glob)):
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if args.kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    names.add(r["Kernel_Name"].split("(")[0])
    if "FETCH_SIZE" not in agg or "WRITE_SIZE" not in agg:
        raise SystemExit("need FETCH_SIZE and WRITE_SIZE passes")
    avg = {c: sum(v) / len(v) for c, v in agg.items()}
    fetch = 2.0 * avg["FETCH_SIZE"] * 1024
    write = avg["WRITE_SIZE"] * 1024
    out = {
        "graph": args.graph, "k": args.k, "order": args.order, "launches_per_spmm": args.launches_per_spmm,
        "source": args.source or args.out,
        "kernel": sorted(names),
        "counters_avg_per_launch": avg,
        "launches_seen": {c: len(v) for c, v in agg.items()},
        "fetch_bytes_per_launch_corrected_x2": fetch,
        "write_bytes_per_launch": write,
        "traffic_bytes_per_launch": fetch + write,
        "algorithmic_bytes_per_launch": args.algorithmic_bytes_per_launch or None,
        "l2_hit_rate": (avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]))
        if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg else None,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 tallies 128-B requests at 64 B); "
                "bytes are L2-miss traffic, Infinity-Cache hits included, so true HBM bytes are lower for "
                "tables that fit the 256 MiB Infinity Cache.",
    }
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))
    if args.merge_into:
        try:
            d = json.load(open(args.merge_into))
        except (OSError, ValueError):
            d = {}
        entries = d.get("entries", [d] if d.get("graph") else [])
        entries = [e for e in entries if (e.get("graph"), e.get("k"), e.get("order", "none")) != (args.graph, args.k, args.order)]
        entries.append(out)
        json.dump({"entries": entries}, open(args.merge_into, "w"), indent=1)


if __name__ == "__main__":
    main()
