#!/bin/bash
# LDS-side PMC pass over the FilterInterpolation C=196 launch for several development flag sets:
#   tools/pmc_fi_lds.sh <tag> <flow> "<flags> <flags> ..."      (through gpurun, needs <pkg>/lib_dev)
set -o pipefail
TAG=${1:-x}; FLOW=${2:-smooth}; FLAGSETS=${3:-"8 72"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$(ls -d $R/*_amd)/lib_dev/libvfi_hip.so
OUT=$R/gpurun_out/pmcfilds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fl in $FLAGSETS; do
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/f$fl -- python3 $R/tools/prof_fi.py --lib $LIB $FLOW 196 $fl > $OUT/f$fl.log 2>&1 || echo "flags $fl failed" >> $OUT/failed
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/f$fl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fi_forward_ori_lds" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("flags $fl $FLOW:", {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
done
