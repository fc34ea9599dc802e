"""Build check: the hand-scheduled GEMM kernels must not use scratch memory (an operand
register that the compiler keeps in scratch is copied while the inline-asm load that fills
it is still in flight -- DESIGN.md, lesson 10).  Reads the -Rpass-analysis=kernel-resource-
usage remarks of one translation unit: python3 check_scratch.py <log> <kernel-name-regex>"""
import re, sys
log, pat = open(sys.argv[1]).read(), re.compile(sys.argv[2])
name, bad = None, []
for line in log.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
    if m and name and pat.search(name) and int(m.group(1)) != 0:
        bad.append((name, int(m.group(1))))
for n, b in bad:
    print("check_scratch: kernel %s uses %d bytes of scratch per lane" % (n, b), file=sys.stderr)
sys.exit(1 if bad else 0)
