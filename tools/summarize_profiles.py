#!/usr/bin/env python3
"""Turns what tools/collect_profiles.sh left under gpurun_out/profiles_<tag>/ into the committed summaries
under profiles/:  python tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    # gpurun merges every run's output into gpurun_out/: take the newest file, not the first name
    hits = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[-1]


# 1. kernel stats of the profiled bench run
shutil.copy(one("bench/*/*kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
with open(os.path.join(src, "bench_under_rocprof.json")) as fh:
    line = [l for l in fh if l.startswith("{")][-1]
with open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w") as fh:
    fh.write(line)

# 2. the same trace split by launch geometry (C=196 vs C=3 launches of the same kernel)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(one("bench/*/*kernel_trace.csv"))):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(dst, tag + "_bench_kernel_by_shape.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "workgroup_x", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow(list(key) + [len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2),
                                round(sum(v) / 1e6, 3)])

# 3. PMC means per launch of the C=196 FilterInterpolation kernel
pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    files = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]   # newest run only
    if not files:
        continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "fi_forward_ori_lds" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc[os.path.basename(d)] = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
with open(os.path.join(dst, tag + "_fi196_pmc.json"), "w") as fh:
    json.dump(pmc, fh, indent=1, sort_keys=True)

# 4. HBM bytes per launch (smooth flow): FETCH_SIZE / WRITE_SIZE are in KB; the window reads (LDS-DMA) count
#    at ~1, the coalesced flow / filter reads (72 B/px, once per channel group) at 1/2 -- profiles/README.md
px = 1152 * 1984
groups = 2
fetch = pmc["pmc_smooth_FETCH_SIZE"]["FETCH_SIZE"] * 1024.0
write = pmc["pmc_smooth_WRITE_SIZE"]["WRITE_SIZE"] * 1024.0
half_counted = 72.0 * px * groups / 2.0
traffic = {"hbm_bytes_per_launch": fetch + half_counted + write, "fetch_bytes_counted": fetch,
           "flow_filter_bytes_half_counted_correction": half_counted, "write_bytes": write,
           "algorithmic_bytes": 1640.0 * px, "source": tag + "_fi196_pmc.json"}
with open(os.path.join(dst, "fi196_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(json.dumps(traffic))
