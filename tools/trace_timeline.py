"""Timeline of ONE replayed step from a rocprofv3 --kernel-trace csv: which kernels overlap (side
stream), the gaps, and what the step's end waits for.

    python tools/trace_timeline.py <dir-or-kernel_trace.csv> [step_marker_kernel=adam_kernel] [which=-2]
prints the kernels between two consecutive launches of the marker kernel (the optimiser launch
ends a step) as  start_us  dur_us  queue  name  [overlap with the previous kernel]."""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(src)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"))
                 for r in rows), key=lambda t: t[0])
    ends = [i for i, k in enumerate(ks) if k[2].startswith(marker)]
    a, b = ends[which - 1] + 1, ends[which] + 1
    step = ks[a:b]
    t0 = ks[a - 1][1]                                # the previous step's optimiser launch ends
    print("step of %d kernels, %.1f us from the previous %s's end to this one's end" %
          (len(step), (step[-1][1] - t0) / 1e3, marker))
    busy_end = t0
    qs = sorted(set(k[3] for k in step))
    per_q = {q: 0 for q in qs}
    for s, e, name, q in step:
        per_q[q] += e - s
        gap = (s - busy_end) / 1e3
        print("%9.1f %8.1f  q%-3s %-70s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:70],
                                                 ("gap %.1f" % gap) if gap > 0 else ("overlap %.1f" % -gap)))
        busy_end = max(busy_end, e)
    for q in qs:
        print("queue %s: %.1f us of kernels" % (q, per_q[q] / 1e3))


if __name__ == "__main__":
    main()
