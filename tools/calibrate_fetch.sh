#!/bin/bash
# tools/calibrate_fetch.sh <tag>   (through gpurun, from the repo root): FETCH_SIZE / EA read-request factors
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetchcal_$TAG
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/probes/fetch_size_calibration.hip -o /tmp/fetchcal || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- /tmp/fetchcal > $OUT/cal_fetch.log 2>&1 || exit 1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/cal_rdreq -- /tmp/fetchcal > $OUT/cal_rdreq.log 2>&1 || exit 1
for model in smooth quarter zero invalid; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fi_fetch_$model -- python3 $R/tools/prof_fi.py $model 196 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/fi_write_$model -- python3 $R/tools/prof_fi.py $model 196 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/fi_rdreq_$model -- python3 $R/tools/prof_fi.py $model 196 > /dev/null 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, json
def read(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
true_bytes = 196 * 1152 * 1984 * 4
res = {"true_bytes": true_bytes, "calibration": {}, "fi196": {}}
f, q = read("$OUT/cal_fetch"), read("$OUT/cal_rdreq")
for k in sorted(f):
    fs = f[k].get("FETCH_SIZE", 0.0) * 1024.0
    rd, rd32 = q.get(k, {}).get("TCC_EA0_RDREQ_sum", 0.0), q.get(k, {}).get("TCC_EA0_RDREQ_32B_sum", 0.0)
    res["calibration"][k] = {"FETCH_SIZE_bytes": fs, "fetch_over_true": fs / true_bytes, "EA_RDREQ": rd, "EA_RDREQ_32B": rd32,
                             "bytes_if_64B_and_32B_requests": (rd - rd32) * 64 + rd32 * 32}
    print("%-28s FETCH_SIZE/true = %.3f   RDREQ %.3e (32B: %.3e)  -> (RDREQ-32B)*64+32B*32 = %.3f of true; RDREQ*128 = %.3f of true"
          % (k, fs / true_bytes, rd, rd32, ((rd - rd32) * 64 + rd32 * 32) / true_bytes, rd * 128 / true_bytes))
for model in ("smooth", "quarter", "zero", "invalid"):
    a, b, c = read("$OUT/fi_fetch_" + model), read("$OUT/fi_write_" + model), read("$OUT/fi_rdreq_" + model)
    for k in a:
        if "fi_forward_ori" not in k:
            continue
        res["fi196"][model] = {"kernel": k, "FETCH_SIZE_bytes": a[k]["FETCH_SIZE"] * 1024.0,
                               "WRITE_SIZE_bytes": b.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0,
                               "EA_RDREQ": c.get(k, {}).get("TCC_EA0_RDREQ_sum", 0.0), "EA_RDREQ_32B": c.get(k, {}).get("TCC_EA0_RDREQ_32B_sum", 0.0)}
        print(model, res["fi196"][model])
json.dump(res, open("$OUT/fetch_calibration.json", "w"), indent=1)
PY
