#!/usr/bin/env python3
"""Condense gpurun_out/{prof,pmc_fetch,pmc_write}_<tag>_<config>_<variant> into profiles/.  Usage:
   python scripts/summarize_profiles.py r03 c3 complete [suffix]"""
import collections, csv, json, os, shutil, sys
tag, cfg, var = (sys.argv[1:4] + ["r03", "c3", "complete"][len(sys.argv) - 1:])[:3]
suffix = sys.argv[4] if len(sys.argv) > 4 else ""
name = "%s_%s_%s%s" % (tag, cfg, var, suffix)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(root, "gpurun_out", "prof_%s" % name, "%s_kernel_stats.csv" % name),
            os.path.join(out, "%s_kernel_stats.csv" % name))
lean = os.path.join(root, "gpurun_out", "prof_%s_lean" % name, "%s_lean_kernel_stats.csv" % name)
if os.path.exists(lean):          # the W+V-steps-only population (bench.py --lean): what roofline.rocprof_avg_us reads
    shutil.copy(lean, os.path.join(out, "%s_lean_kernel_stats.csv" % name))
clean = os.path.join(root, "gpurun_out", "bench_%s.log" % name)      # the un-profiled run of the same command, when the job made one
log = open(clean if os.path.exists(clean) else os.path.join(root, "gpurun_out", "prof_%s.log" % name)).read().splitlines()
line = [ln for ln in log if ln.startswith("{") and '"metric"' in ln]
if line:
    open(os.path.join(out, "%s_bench.json" % name), "w").write(line[-1] + "\n")
pmc = {}
for nm, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (nm, name), "%s_counter_collection.csv" % name))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[ctr + "_KB_avg"] = sum(v) / len(v)
        pmc[k]["launches_" + nm] = len(v)
sqdir = os.path.join(root, "gpurun_out", "pmc_sq_%s" % name, "%s_counter_collection.csv" % name)
if os.path.exists(sqdir):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sqdir)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for cn, v in d.items():
            pmc.setdefault(k, {})[cn + "_avg"] = sum(v) / len(v)
# gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM)
for k, d in pmc.items():
    f, w = d.get("FETCH_SIZE_KB_avg", 0.0), d.get("WRITE_SIZE_KB_avg", 0.0)
    d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
json.dump(pmc, open(os.path.join(out, "%s_pmc_%s_%s%s.json" % (tag, cfg, var, suffix)), "w"), indent=1, sort_keys=True)
for k, d in sorted(pmc.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0.0))[:6]:
    print("%-60s %10.2f MB/launch" % (k[:60], d["hbm_bytes_per_launch_corrected"] / 1e6))
