#!/bin/bash
# PMC passes over two secondary kernels, separate --pmc runs with --kernel-trace only, and the
# timings of the ops that are not in the bench step:   tools/collect_extras.sh <tag>   (repo root, through gpurun)
#   -> gpurun_out/extras_<tag>/{f16,corr}_pmc.txt, rest_ops.txt
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/extras_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for what in f16 corr; do
  i=0
  for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${what}_p$i -- python3 $R/tools/prof_misc.py $what > $OUT/${what}_p$i.log 2>&1 || exit 1
  done
  echo $what-pmc-done
done
python3 $R/tools/bench_rest.py > $OUT/rest_ops.txt 2>/dev/null || exit 1
python3 $R/tools/bench_bwd.py >> $OUT/rest_ops.txt 2>/dev/null || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = {}
for what, part in (("f16", "fi_forward_ori_lds_f16"), ("corr", "corr_forward")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s_p*/**/*counter_collection.csv" % (out, what), recursive=True):
        for r in csv.DictReader(open(f)):
            if part in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob("%s/%s_p1/**/*kernel_trace.csv" % (out, what), recursive=True):
        for r in csv.DictReader(open(f)):
            if part in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res[what] = {}
    for k, d in agg.items():
        e = {c: sum(v) / len(v) for c, v in sorted(d.items())}
        e["read_bytes (EA_RDREQ x 128 B)"] = e.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0
        e["write_bytes (WRITE_SIZE x 1024)"] = e.get("WRITE_SIZE", 0.0) * 1024.0
        if dur.get(k): e["avg_us (counter pass)"] = sum(dur[k]) / len(dur[k])
        res[what][k] = e
json.dump(res, open(out + "/f16_corr_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({w: {k: {c: v[c] for c in v if "bytes" in c or "avg_us" in c} for k, v in d.items()} for w, d in res.items()}, indent=1))
PY
echo collected > $OUT/done
