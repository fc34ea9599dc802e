"""MFMA utilisation per kernel from a rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \\
              --output-format csv -d <dir> -- python3 bench.py --no-graph ...
    python tools/pmc_mfma.py <dir> <steps> [algorithmic GFLOP per step]
prints csv rows kernel,dispatches_per_step,us_per_step,mfma_busy_frac,executed_gflop_per_step.

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * kernel cycles), kernel
cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs,
MI355X_MICROARCH.md "DVFS give-back"): the share of all matrix-pipe cycles of the chip, at
the clock the chip actually held, in which an MFMA was executing.  executed GFLOP =
SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 (counts padded rows / columns too, so executed /
algorithmic is the padding overhead)."""
import csv, glob, os, sys
d, steps = sys.argv[1], int(sys.argv[2])
alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc.setdefault(r["Kernel_Name"], {"n": {}, "busy": 0.0, "mops": 0.0, "act": 0.0, "ns": 0.0})
        c, v = r["Counter_Name"], float(r["Counter_Value"])
        if c == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["busy"] += v
            a["n"][r["Dispatch_Id"]] = 1
            a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        elif c == "SQ_INSTS_VALU_MFMA_MOPS_F32":
            a["mops"] += v
        elif c == "GRBM_GUI_ACTIVE":
            a["act"] += v
print("kernel,dispatches_per_step,us_per_step,mfma_busy_frac,executed_gflop_per_step")
tb = tm = ta = tn = 0.0
for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
    cyc = a["act"] / 8.0
    frac = a["busy"] / (1024.0 * cyc) if cyc > 0 else 0.0
    print('"%s",%.2f,%.2f,%.4f,%.3f' % (k, len(a["n"]) / steps, a["ns"] / steps / 1e3, frac,
                                       a["mops"] * 512 / steps / 1e9))
    tb += a["busy"]; tm += a["mops"]; ta += cyc; tn += a["ns"]
print('"TOTAL (whole step)",,%.2f,%.4f,%.3f' % (tn / steps / 1e3, tb / (1024.0 * ta) if ta else 0.0,
                                              tm * 512 / steps / 1e9))
if alg:
    print('"executed / algorithmic FLOP",,,,%.4f' % (tm * 512 / steps / 1e9 / alg))
