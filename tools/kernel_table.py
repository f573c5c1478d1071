#!/usr/bin/env python3
"""Per-kernel table of a profiled workload: average duration (rocprofv3 kernel stats), algorithmic bytes
(bench.py's ALGO_ARRAYS), achieved GB/s and share of the 8 TB/s peak, HBM traffic from the PMC passes over the
algorithmic bytes.  usage: tools/kernel_table.py <round> <stats tag> <traffic tag> <Lm> <Mm> <N>"""
import csv, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_pmc import canon

rnd, stag, ttag, Lm, Mm, N = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
tr = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_{ttag}_traffic.json")))["kernels"]
rows = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{rnd}_{stag}_kernel_stats.csv"))):
    k = canon(r["Name"])
    c, t = rows.get(k, (0, 0.0))
    rows[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
print("| kernel | avg µs | algorithmic MB | GB/s | of 8 TB/s | HBM traffic / algorithmic | pair |")
print("|---|---|---|---|---|---|---|")
pair = 0.0
for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    ab = bench.algo_bytes(k, Lm, Mm, N)
    if ab is None:
        continue
    us = t / c / 1e3
    hb = tr.get(k, {}).get("hbm_bytes_per_launch")
    inpair = k in bench.PAIR_KERNELS
    if inpair:
        pair += us * (2 if False else 1)
    print(f"| `{k}` | {us:.1f} | {ab/1e6:.1f} | {ab/us/1e3:.0f} | {ab/us/1e3/80:.1f} % | {hb/ab:.2f} | {'x' if inpair else ''} |" if hb else
          f"| `{k}` | {us:.1f} | {ab/1e6:.1f} | {ab/us/1e3:.0f} | {ab/us/1e3/80:.1f} % | - | {'x' if inpair else ''} |")
print(f"\npair kernels, sum of averages: {pair:.0f} µs")
