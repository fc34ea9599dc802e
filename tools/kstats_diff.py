"""Per-kernel regression gate: a new rocprofv3 `*_kernel_stats.csv` of a bench workload against
the profile of record.

    python tools/kstats_diff.py <new kernel_stats.csv> [<old kernel_stats.csv> | --workload W]
                                [--steps 25] [--pct 3] [--us 2]

Both files come from `bench.py --workload W --steps 20 --warmup 5` under `rocprofv3
--kernel-trace --stats` (tools/gpu_round.sh); the number of steps in a file is the call count of
the optimiser launch (25 until round 5, more since the set-up of the multi-step graphs): a kernel's figure is its TOTAL
duration per step (all its launches; a kernel that is launched more or less often than before
is compared as a whole).  Exit status 1 if any kernel that exists in both grew by more than
--pct percent AND more than --us microseconds per step, or if the step's kernel sum grew by
more than 1 %.  New / vanished kernels are listed; they count through the sum.

Why it exists: round 3 shipped +16 us per step in the first-layer backward (run-time debug
branches, occupancy 3 -> 2) and +13 us in the dominant GEMM (a re-tune never A/B-ed inside the
step) while its A/B experiments were measuring +-2 us effects elsewhere.
`tools/adopt_profile.py` runs this before `profiles/CURRENT.json` moves."""
import argparse
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def steps_of(path, default):
    """the optimiser launch runs once per step: its call count is the number of steps in the file
    (20 + 5 until round 5; since bench.py runs several steps per graph launch the set-up adds more)"""
    for r in csv.DictReader(open(path)):
        if short(r["Name"]) in ("adam_kernel", "sgd_kernel"):
            return int(r["Calls"])
    return default


def load(path, steps):
    steps = steps_of(path, steps)
    out = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        us, calls = out.get(k, (0.0, 0.0))
        out[k] = (us + float(r["TotalDurationNs"]) / 1e3 / steps, calls + int(r["Calls"]) / steps)
    return out


def record_path(workload):
    cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
    tag = cur[workload]["tag"]
    return os.path.join(ROOT, "profiles", "%s_bench_%s_kernel_stats.csv" % (tag, workload))


def diff(new_path, old_path, steps=25, pct=3.0, us=2.0, out=sys.stdout):
    new, old = load(new_path, steps), load(old_path, steps)
    bad = []
    rows = []
    for k in sorted(set(new) | set(old), key=lambda k: -max(new.get(k, (0, 0))[0], old.get(k, (0, 0))[0])):
        n, o = new.get(k), old.get(k)
        if n and o:
            d = n[0] - o[0]
            flag = ""
            if d > us and d > o[0] * pct / 100.0:
                flag = "REGRESSION"
                bad.append(k)
            elif -d > us and -d > o[0] * pct / 100.0:
                flag = "faster"
            rows.append((k, o[0], n[0], d, o[1], n[1], flag))
        elif n:
            rows.append((k, 0.0, n[0], n[0], 0.0, n[1], "new"))
        else:
            rows.append((k, o[0], 0.0, -o[0], o[1], 0.0, "gone"))
    tn, to = sum(v[0] for v in new.values()), sum(v[0] for v in old.values())
    print("%-52s %9s %9s %8s  %s" % ("kernel (us per step, all launches)", "record", "new", "delta", "launches/step"), file=out)
    for k, o, n, d, oc, nc, flag in rows:
        if abs(d) < 0.5 and not flag:
            continue
        print("%-52s %9.1f %9.1f %+8.1f  %g -> %g %s" % (k[:52], o, n, d, oc, nc, flag), file=out)
    print("%-52s %9.1f %9.1f %+8.1f  (%+.2f %%)" % ("SUM of the step's kernels", to, tn, tn - to, 100.0 * (tn - to) / to), file=out)
    if tn > to * 1.01:
        bad.append("SUM")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("new")
    ap.add_argument("old", nargs="?")
    ap.add_argument("--workload")
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--pct", type=float, default=3.0)
    ap.add_argument("--us", type=float, default=2.0)
    a = ap.parse_args()
    old = a.old or record_path(a.workload)
    print("record: %s\nnew:    %s" % (os.path.relpath(old, ROOT), a.new))
    bad = diff(a.new, old, a.steps, a.pct, a.us)
    if bad:
        print("kstats_diff: REGRESSED: %s" % ", ".join(bad))
        sys.exit(1)
    print("kstats_diff: ok")


if __name__ == "__main__":
    main()
