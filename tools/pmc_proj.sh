#!/bin/bash
# PMC passes over the projection kernels: tools/pmc_proj.sh <tag> [flow]   (through gpurun, repo root)
set -o pipefail
TAG=${1:-x}; FLOW=${2:-smooth}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcproj_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_proj.py --flows $FLOW --iters 20 > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("    %-26s n=%4d mean %14.1f" % (c, len(v), sum(v) / len(v)))
PY
