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
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
H, W = 1152, 1984
PX = H * W


def one(pattern):
    # gpurun merges every run's output into gpurun_out/: take the newest file, not the first name
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[-1]


def by_shape(trace, out):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "workgroup_x", "calls", "avg_us", "min_us", "max_us", "total_ms"])
        for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow(list(key) + [len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2),
                                    round(sum(v) / 1e6, 3)])


def counters(d, kernel_substr):
    f = one(os.path.join(d, "**", "*counter_collection.csv"))
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"]:
            vals[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {"%s %s" % k: sum(v) / len(v) for k, v in sorted(vals.items())}


# 1. kernel stats + by-shape split of the profiled bench run (the driver's command)
shutil.copy(one("bench/**/*kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
with open(os.path.join(src, "bench_under_rocprof.json")) as fh:
    line = [l for l in fh if l.startswith("{")][-1]
with open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w") as fh:
    fh.write(line)
by_shape(one("bench/**/*kernel_trace.csv"), os.path.join(dst, tag + "_bench_kernel_by_shape.csv"))
by_shape(one("proj/**/*kernel_trace.csv"), os.path.join(dst, tag + "_proj_kernel_by_shape.csv"))

# 2. PMC means per launch of the C=196 FilterInterpolation kernel
pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    pmc[os.path.basename(d)] = counters(d, "fi_forward_ori_lds")
with open(os.path.join(dst, tag + "_fi196_pmc.json"), "w") as fh:
    json.dump(pmc, fh, indent=1, sort_keys=True)

# 3. the calibration: every load flavour reads the same 1.79 GB exactly once
true_bytes = 196 * PX * 4
cal = {"true_bytes_per_launch": true_bytes, "kernels": {}}
cf, cr = counters(os.path.join(src, "cal_FETCH_SIZE"), "calib_"), counters(os.path.join(src, "cal_RDREQ"), "calib_")
for k, v in cf.items():
    name = k.split(" ")[0]
    rd = cr.get(name + " TCC_EA0_RDREQ_sum", 0.0)
    cal["kernels"][name] = {"FETCH_SIZE_bytes_over_true": round(v * 1024.0 / true_bytes, 4), "EA_RDREQ": rd,
                            "EA_RDREQ_32B": cr.get(name + " TCC_EA0_RDREQ_32B_sum", 0.0),
                            "RDREQ_x_128B_over_true": round(rd * 128.0 / true_bytes, 4)}
with open(os.path.join(dst, tag + "_fetch_calibration.json"), "w") as fh:
    json.dump(cal, fh, indent=1, sort_keys=True)

# 4. HBM-side bytes per launch: reads = EA read requests x 128 B (the calibration: every flavour, the LDS-DMA window
#    loads included, issues 128-byte requests that FETCH_SIZE tallies at 64), writes = WRITE_SIZE (KB, exact)
entries = []
for model in ("smooth", "quarter"):
    kn = [k for k in pmc["pmc_%s_TCC_EA0_RDREQ_sum" % model] if k.endswith("TCC_EA0_RDREQ_sum")][0]
    rd = pmc["pmc_%s_TCC_EA0_RDREQ_sum" % model][kn] * 128.0
    wk = [k for k in pmc["pmc_%s_WRITE_SIZE" % model] if k.endswith("WRITE_SIZE")][0]
    wr = pmc["pmc_%s_WRITE_SIZE" % model][wk] * 1024.0
    entries.append({"h": H, "w": W, "flow_model": model, "direct": False, "kernel": kn.rsplit(" ", 1)[0],
                    "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                    "algorithmic_bytes": 1640.0 * PX, "ratio": round((rd + wr) / (1640.0 * PX), 3),
                    "source": "profiles/%s_fi196_pmc.json (EA_RDREQ x 128 B + WRITE_SIZE)" % tag})

# 5. the north-star gate's kernels: per-launch PMC means of the FilterInterpolation C=3 launch and of the three launches of a
#    FlowProjection call (FlowProjection only: the `false` instantiations), and their HBM-side bytes per call
gate = {"fi_c3": {}, "flowproj": {}}
for d in sorted(glob.glob(os.path.join(src, "gate_fi3_*"))):
    for k, v in counters(d, "fi_forward_ori_lds").items():
        gate["fi_c3"][k] = v
for d in sorted(glob.glob(os.path.join(src, "gate_proj_*"))):
    for k, v in counters(d, "proj_").items():
        if "<true" not in k and "proj_backward" not in k:
            gate["flowproj"][k] = v
with open(os.path.join(dst, tag + "_gate_pmc.json"), "w") as fh:
    json.dump(gate, fh, indent=1, sort_keys=True)


def traffic_of(block, name_part):
    rd = sum(v for k, v in block.items() if k.endswith("TCC_EA0_RDREQ_sum") and name_part in k) * 128.0
    wr = sum(v for k, v in block.items() if k.endswith("WRITE_SIZE") and name_part in k) * 1024.0
    return rd, wr


if gate["fi_c3"]:
    rd, wr = traffic_of(gate["fi_c3"], "fi_forward_ori_lds")
    entries.append({"h": H, "w": W, "flow_model": "smooth", "direct": False, "op": "fi_c3", "kernel": "vfi::fi_forward_ori_lds<false, 0> (C=3)",
                    "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes": 96.0 * PX,
                    "ratio": round((rd + wr) / (96.0 * PX), 3), "source": "profiles/%s_gate_pmc.json (EA_RDREQ x 128 B + WRITE_SIZE)" % tag})
if gate["flowproj"]:
    rd, wr = traffic_of(gate["flowproj"], "proj_")
    entries.append({"h": H, "w": W, "flow_model": "smooth", "direct": False, "op": "flowproj",
                    "kernel": "vfi::proj_scan4<false> + proj_pull_lean<false> + proj_finish (one FlowProjection call)",
                    "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes": 20.0 * PX,
                    "ratio": round((rd + wr) / (20.0 * PX), 3), "source": "profiles/%s_gate_pmc.json (EA_RDREQ x 128 B + WRITE_SIZE, summed over the three launches)" % tag})
# (round 3) the fp16-storage C=196 launch and the finest correlation level: tools/collect_extras.sh
extras = os.path.join(os.path.dirname(src), "extras_" + tag)
if os.path.exists(os.path.join(extras, "f16_corr_pmc.json")):
    shutil.copy(os.path.join(extras, "f16_corr_pmc.json"), os.path.join(dst, tag + "_f16_corr_pmc.json"))
    shutil.copy(os.path.join(extras, "rest_ops.txt"), os.path.join(dst, tag + "_rest_ops.txt"))
    with open(os.path.join(extras, "f16_corr_pmc.json")) as fh:
        ex = json.load(fh)
    for op, block, kern, alg in (("fi196_f16", ex.get("f16", {}), "fi_forward_ori_lds_f16", 856.0 * PX),
                                 ("corr_finest", ex.get("corr", {}), "corr_forward_k1_quad", (2 * 32 + 81) * 4.0 * (H // 4) * (W // 4))):
        for k, v in block.items():
            if kern in k:
                rd, wr = v["read_bytes (EA_RDREQ x 128 B)"], v["write_bytes (WRITE_SIZE x 1024)"]
                entries.append({"h": H, "w": W, "flow_model": "smooth", "direct": False, "op": op, "kernel": k,
                                "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes": alg,
                                "ratio": round((rd + wr) / alg, 3), "source": "profiles/%s_f16_corr_pmc.json (EA_RDREQ x 128 B + WRITE_SIZE)" % tag})
# (round 4) the list form of the projections and the shared-window launch
if os.path.isdir(os.path.join(src, "projbatch")):
    by_shape(one("projbatch/**/*kernel_trace.csv"), os.path.join(dst, tag + "_projbatch_kernel_by_shape.csv"))
    shutil.copy(os.path.join(src, "projbatch_bench.log"), os.path.join(dst, tag + "_projbatch_bench.txt"))
    batch = {}
    for d in sorted(glob.glob(os.path.join(src, "batch_proj_*"))):
        f = one(os.path.join(os.path.basename(d), "**", "*counter_collection.csv"))
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "proj_" in r["Kernel_Name"]:
                # split by launch size: the grid tells how many items a launch carried
                key = "%s grid_x=%s %s" % (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"), r["Counter_Name"])
                vals[key].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            batch[k] = sum(v) / len(v)
    with open(os.path.join(dst, tag + "_projbatch_pmc.json"), "w") as fh:
        json.dump(batch, fh, indent=1, sort_keys=True)
    multi = {}
    for d in sorted(glob.glob(os.path.join(src, "multi3_*"))):
        for k, v in counters(d, "fi_forward_ori_multi").items():
            multi[k] = v
    with open(os.path.join(dst, tag + "_multi3_pmc.json"), "w") as fh:
        json.dump(multi, fh, indent=1, sort_keys=True)
with open(os.path.join(dst, "traffic_by_config.json"), "w") as fh:
    json.dump({"entries": entries}, fh, indent=1)
print(json.dumps(entries, indent=1))
