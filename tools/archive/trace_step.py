"""One step's kernel sequence (name, us) from a rocprofv3 kernel trace of a bench run:
usage: python tools/trace_step.py <dir with *kernel_trace.csv> [first-kernel substring]
Prints the LAST complete step (from one launch of the first kernel to the next)."""
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "fill_multi_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"),
                                      r["Kernel_Name"][:90]))
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
