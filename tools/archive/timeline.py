"""print the kernel timeline of the last step from a rocprofv3 --kernel-trace output dir"""
import csv, glob, sys
d = sys.argv[1]
kt = sorted(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('pack_multi')]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot += dur
    nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
    print("%8.1f %7.1f  %s grid=%s vgpr=%s" % (s / 1e3, dur / 1e3, nm[:48], r.get('Grid_Size_X'), r.get('VGPR_Count')))
print("sum of kernels %.1f us, span %.1f us" % (tot / 1e3, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
