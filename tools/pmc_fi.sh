#!/bin/bash
# (a pass with TA_*/TCP_* counters aborted rocprofv3 on this image and is left out)
# PMC passes over the FilterInterpolation C=196 launch: tools/pmc_fi.sh <tag> <flow> <flags> (through gpurun, repo root)
set -o pipefail
TAG=${1:-x}; FLOW=${2:-smooth}; FLAGS=${3:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcfi_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_WAVE32_LDS" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/prof_fi.py $FLOW 196 $FLAGS > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed
done
echo done > $OUT/done
