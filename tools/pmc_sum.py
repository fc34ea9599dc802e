"""Sum a rocprofv3 --pmc counter per kernel: python tools/pmc_sum.py <dir> <counter> <steps>
prints csv rows counter,kernel,dispatches,steps,sum_KB,MB_per_step (FETCH_SIZE / WRITE_SIZE
count KB on gfx950, see MI355X_MICROARCH.md)."""
import csv, glob, os, sys
d, counter, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"]
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
tot = 0.0
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%s,"%s",%d,%d,%r,%r' % (counter, k, n, steps, v, v / steps / 1024.0))
    tot += v
print('%s,"TOTAL",,%d,%r,%r' % (counter, steps, tot, tot / steps / 1024.0))
