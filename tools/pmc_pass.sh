#!/bin/bash
# Generic PMC passes (separate --pmc runs with --kernel-trace only) over a python driver, with a per-kernel summary:
#   tools/pmc_pass.sh <tag> <kernel substring> <driver.py> [driver args...]      (through gpurun, repo root)
set -o pipefail
TAG=$1; KSUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do     # (a TCC_* pass aborted rocprofv3 on this image with these drivers: collected by tools/collect_profiles.sh's own passes instead)
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/"$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed
  echo "pass $i done"
done
python3 - "$OUT" "$KSUB" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("void vfi::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].replace("void vfi::", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in sorted(agg.items()):
    if sys.argv[2] not in k: continue
    print(k, "launches traced %d, mean %.1f us" % (len(dur[k]), sum(dur[k]) / max(1, len(dur[k])) / 1e3))
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"]) if "SQ_WAVES" in d else 1
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print("    %-26s n=%3d mean %16.1f  per wave %12.1f" % (c, len(v), m, m / w))
PY
