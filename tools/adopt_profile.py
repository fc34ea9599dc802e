"""Adopt the output of tools/gpu_round.sh as the profile of record of a workload:
    python tools/adopt_profile.py gpurun_out/<dir> <tag> <workload>
copies bench.json / kernel_stats.csv / pmc_mfma.csv / pmc_fetch.csv + pmc_write.csv (joined as
pmc_traffic.csv) to profiles/<tag>_bench_<workload>*.{json,csv} and points
profiles/CURRENT.json at them, with the identity (sources_sha16) of the kernel sources and
tilings the profile was TAKEN with (written on the GPU box by gpu_round.sh) -- bench.py
reads roofline.traffic / mfma_util from there and reports null once the sources move on."""
import json, os, shutil, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
src, tag, wl = args[0], args[1], args[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tools"))
import kstats_diff
# The per-kernel regression gate: the profile of record only moves to a
# profile in which no kernel got slower by > 3 % and > 2 us per step and the step's kernel sum
# did not grow by > 1 % -- or with `--accept-regression "<why>"`, which is recorded.
cur_path0 = os.path.join(root, "profiles", "CURRENT.json")
accept = None
for i, a in enumerate(sys.argv):
    if a.startswith("--accept-regression"):
        accept = a.split("=", 1)[1] if "=" in a else "unexplained"
if os.path.exists(cur_path0) and wl in json.load(open(cur_path0)):
    old = kstats_diff.record_path(wl)
    bad = kstats_diff.diff(os.path.join(src, "kernel_stats.csv"), old)
    if bad and accept is None:
        sys.exit("adopt_profile: per-kernel regression against %s: %s\n"
                 "   (fix it, or pass --accept-regression=<why>)" % (os.path.basename(old), ", ".join(bad)))
else:
    bad = []
base = os.path.join(root, "profiles", "%s_bench_%s" % (tag, wl))
shutil.copy(os.path.join(src, "bench.json"), base + ".json")
shutil.copy(os.path.join(src, "kernel_stats.csv"), base + "_kernel_stats.csv")
shutil.copy(os.path.join(src, "pmc_mfma.csv"), base + "_pmc_mfma.csv")
with open(base + "_pmc_traffic.csv", "w") as f:
    f.write(open(os.path.join(src, "pmc_fetch.csv")).read())
    f.write(open(os.path.join(src, "pmc_write.csv")).read())
sha = open(os.path.join(src, "sources_sha16.txt")).read().strip()
cur_path = os.path.join(root, "profiles", "CURRENT.json")
cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
cur[wl] = {"tag": tag, "sources_sha16": sha}
if bad:
    cur[wl]["accepted_regression"] = {"kernels": bad, "why": accept}
json.dump(cur, open(cur_path, "w"), indent=1, sort_keys=True)
print("profiles/%s_bench_%s* adopted (sources %s)" % (tag, wl, sha))
