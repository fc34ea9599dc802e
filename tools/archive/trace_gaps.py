"""Gaps between consecutive kernels of the graph-replayed step, from a rocprofv3 kernel trace:
usage: python tools/trace_gaps.py <dir with *kernel_trace.csv> [skip_steps]
Prints, for the steady-state part of the trace: kernels per step, sum of durations, sum of
gaps (start of a kernel - end of the one before it), and the largest gaps by kernel pair."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last 40 % of the dispatches
rows = rows[int(len(rows) * 0.6):]
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = defaultdict(lambda: [0, 0])
tot_gap = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    tot_gap += g
    k = (a["Kernel_Name"][:40], b["Kernel_Name"][:40])
    gaps[k][0] += g
    gaps[k][1] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("dispatches %d  span %.1f us  durations %.1f us (%.1f %%)  gaps %.1f us (%.1f %%), mean gap %.2f us"
      % (len(rows), span / 1e3, dur / 1e3, 100.0 * dur / span, tot_gap / 1e3, 100.0 * tot_gap / span,
         tot_gap / 1e3 / (len(rows) - 1)))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %8.2f us mean over %3d   %s -> %s" % (g / n / 1e3, n, k[0], k[1]))
