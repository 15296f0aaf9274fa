#!/bin/bash
# experiment: tile -> XCD grouping of fi_forward_ori_lds; time and EA read requests per launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fi_xcd_exp
mkdir -p $OUT
python3 $R/tools/bench_ops.py --ops fi196,fi3 --flows smooth --knobs 0:0,8:0,16:0,32:0 > $OUT/time.log 2>&1
grep knob $OUT/time.log
cd /tmp && export TMPDIR=/tmp
for fl in 16 32; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/p$fl -- python3 $R/tools/prof_fi.py smooth 196 $fl > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p$fl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fi_forward_ori_lds" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("flags $fl:", {k: "%.3e" % (sum(v) / len(v)) for k, v in agg.items()}, " read GB = %.2f" % (sum(agg["TCC_EA0_RDREQ_sum"]) / len(agg["TCC_EA0_RDREQ_sum"]) * 128 / 1e9))
PY
done
