"""Build check on the -Rpass-analysis=kernel-resource-usage remarks of one translation unit:
    python3 check_scratch.py <log> <kernel-name-regex> [<budget file>]
1. The hand-scheduled GEMM kernels (names matching the regex) must not use scratch memory: an
   operand register that the compiler keeps in scratch is copied while the inline-asm load that
   fills it is still in flight (lesson 10).
2. Kernels whose launch geometry assumes an occupancy (persistent grids of "G work-groups per
   CU", two co-resident work-groups) must keep it: the budget file holds lines
   `<regex over the mangled name> <min waves/SIMD> [<max VGPRs>]`; a kernel that matches and
   falls below fails the build.  (Round 3 shipped run-time debug branches that took the
   first-layer backward from 103 to 128 VGPRs = occupancy 3 -> 2 under a 3-per-CU grid:
   +16 us per step, unnoticed.)"""
import re, sys
log, pat = open(sys.argv[1]).read(), re.compile(sys.argv[2])
budget = []
if len(sys.argv) > 3:
    for line in open(sys.argv[3]):
        line = line.split('#')[0].split()
        if line:
            budget.append((re.compile(line[0]), int(line[1]), int(line[2]) if len(line) > 2 else None))
name, bad = None, []
vg = None
for line in log.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
    if m and name and pat.search(name) and int(m.group(1)) != 0:
        bad.append("kernel %s uses %d bytes of scratch per lane" % (name, int(m.group(1))))
    m = re.search(r"remark:\s+VGPRs: (\d+)", line)
    if m:
        vg = int(m.group(1))
    m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
    if m and name:
        for rx, occ, maxv in budget:
            if rx.search(name):
                if int(m.group(1)) < occ:
                    bad.append("kernel %s: occupancy %d waves/SIMD, its launch geometry needs %d"
                               % (name, int(m.group(1)), occ))
                if maxv is not None and vg is not None and vg > maxv:
                    bad.append("kernel %s: %d VGPRs, budget %d" % (name, vg, maxv))
for b in bad:
    print("check_scratch: " + b, file=sys.stderr)
sys.exit(1 if bad else 0)
