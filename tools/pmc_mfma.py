"""MFMA utilisation per kernel from a rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \\
              --output-format csv -d <dir> -- python3 bench.py --no-graph ...
    python tools/pmc_mfma.py <dir> <steps> [algorithmic GFLOP per step]
prints csv rows kernel,dispatches_per_step,us_per_step,mfma_busy_frac,executed_gflop_per_step,busy_frac_grbm.

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * kernel time * 2.4 GHz): the
share of the chip's matrix-pipe cycles AT THE PEAK CLOCK in which an MFMA was executing --
equal to executed FLOPs / peak FLOPs for f32 MFMAs, whatever clock the chip held.
busy_frac_grbm = the same cycles / (1024 * GRBM_GUI_ACTIVE / 8), round 2's definition:
rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs and the quotient reads HIGH on dispatches
shorter than ~0.3 ms (MI355X_MICROARCH.md "DVFS give-back"; tools/clock_check.py: 2.47-2.55
"GHz" where the in-kernel clock is 2.35-2.38), so this column under-reports by 4-7 %.
executed GFLOP = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 (counts padded rows / columns too, so
executed / algorithmic is the padding overhead)."""
import csv, glob, os, sys
d, steps = sys.argv[1], int(sys.argv[2])
alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
PEAK_GHZ = 2.4
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
print("kernel,dispatches_per_step,us_per_step,mfma_busy_frac,executed_gflop_per_step,busy_frac_grbm")
tb = tm = ta = tn = 0.0
for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
    cyc = a["act"] / 8.0
    frac_g = a["busy"] / (1024.0 * cyc) if cyc > 0 else 0.0
    frac = a["busy"] / (1024.0 * a["ns"] * PEAK_GHZ) if a["ns"] > 0 else 0.0
    print('"%s",%.2f,%.2f,%.4f,%.3f,%.4f' % (k, len(a["n"]) / steps, a["ns"] / steps / 1e3, frac,
                                            a["mops"] * 512 / steps / 1e9, frac_g))
    tb += a["busy"]; tm += a["mops"]; ta += cyc; tn += a["ns"]
print('"TOTAL (whole step)",,%.2f,%.4f,%.3f,%.4f' % (tn / steps / 1e3, tb / (1024.0 * tn * PEAK_GHZ) if tn else 0.0,
                                                   tm * 512 / steps / 1e9, tb / (1024.0 * ta) if ta else 0.0))
if alg:
    print('"executed / algorithmic FLOP",,,,%.4f,' % (tm * 512 / steps / 1e9 / alg))
