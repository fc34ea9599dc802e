"""Adopt the output of tools/gpu_round.sh as the profile of record of a workload:
    python tools/adopt_profile.py gpurun_out/<dir> <tag> <workload>
copies bench.json / kernel_stats.csv / pmc_mfma.csv / pmc_fetch.csv + pmc_write.csv (joined as
pmc_traffic.csv) to profiles/<tag>_bench_<workload>*.{json,csv} and points
profiles/CURRENT.json at them, with the identity (sources_sha16) of the kernel sources and
tilings the profile was TAKEN with (written on the GPU box by gpu_round.sh) -- bench.py
reads roofline.traffic / mfma_util from there and reports null once the sources move on."""
import json, os, shutil, sys
src, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
json.dump(cur, open(cur_path, "w"), indent=1, sort_keys=True)
print("profiles/%s_bench_%s* adopted (sources %s)" % (tag, wl, sha))
