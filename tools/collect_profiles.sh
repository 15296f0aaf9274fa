#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box:
#   tools/collect_profiles.sh <tag>      (run from the repo root through gpurun)
# 1. kernel trace + stats of the default bench command
# 2. PMC passes (separate runs, --kernel-trace only): FETCH_SIZE, WRITE_SIZE, L2 hit/miss, LDS / wave counters
#    for the FilterInterpolation C=196 launch, plus a FETCH_SIZE calibration on a launch with known traffic
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  n=$(echo $ctr | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_smooth_$n -- python3 $R/tools/prof_fi.py smooth 196 > /dev/null 2>&1 || exit 1
done
for model in invalid zero quarter; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${model}_$ctr -- python3 $R/tools/prof_fi.py $model 196 > /dev/null 2>&1 || exit 1
  done
done
echo collected > $OUT/done
