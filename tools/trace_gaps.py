#!/usr/bin/env python3
"""GPU busy / idle time from a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]

Prints, for the second half of the trace (steady state), the wall time covered by at least one kernel,
the idle time between kernels, and per kernel the summed duration and the idle time that precedes it."""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
rows = rows[len(rows) // 2:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; gap_before = defaultdict(float); dur = defaultdict(float); cnt = defaultdict(int)
cur_end = rows[0][0]
for s, e, n in rows:
    if s > cur_end:
        gap_before[n] += s - cur_end
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
    dur[n] += e - s; cnt[n] += 1
wall = t1 - t0
print("wall %.1f us, busy %.1f us (%.1f%%), idle %.1f us, launches %d" % (wall / 1e3, busy / 1e3, 100.0 * busy / wall, (wall - busy) / 1e3, len(rows)))
for n in sorted(dur, key=lambda k: -dur[k])[:25]:
    print("  %-22s n=%5d dur %9.1f us (%.2f us each)  idle before %8.1f us (%.2f each)" % (n[:22], cnt[n], dur[n] / 1e3, dur[n] / 1e3 / cnt[n], gap_before[n] / 1e3, gap_before[n] / 1e3 / cnt[n]))
print("largest idle-before totals:")
for n in sorted(gap_before, key=lambda k: -gap_before[k])[:12]:
    print("  %-22s idle before %8.1f us (%.2f each, n=%d)" % (n[:22], gap_before[n] / 1e3, gap_before[n] / 1e3 / cnt[n], cnt[n]))
