#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box:
#   tools/collect_profiles.sh <tag>      (run from the repo root through gpurun)
# 1. kernel trace + stats of the default bench command (the command the driver runs)
# 2. PMC passes over the dominant launch (FilterInterpolation, C=196, 1152x1984), separate --pmc runs with --kernel-trace
#    only: EA read requests / FETCH_SIZE / WRITE_SIZE / L2 hits / LDS and wave counters, for the smooth and the quarter
#    flow field and two launches of known traffic (invalid = pure copy-through, zero = zero flow)
# 3. the FETCH_SIZE calibration probe (tools/probes/fetch_size_calibration.hip): every load flavour on known bytes
# 4. kernel trace of the projection bench (tools/bench_proj.py): the three launches of a call
# 5. (round 3) PMC passes over the north-star gate's kernels: the FilterInterpolation C=3 launch and the three launches
#    of a FlowProjection call -- EA read requests, WRITE_SIZE, instruction and wait counters (tools/prof_fi.py smooth 3,
#    tools/bench_proj.py)
# 6. (round 4) the list form of the projections: kernel trace of tools/bench_r4.py --what projbatch (n = 1, 2, 6 items per launch
#    triple) and PMC passes over the same driver; the shared-window FilterInterpolation launch: PMC passes over tools/prof_multi.py
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/probes/fetch_size_calibration.hip -o /tmp/fetchcal || exit 1
cd /tmp && export TMPDIR=/tmp
# --no-extras: only the timed region, so that the dominant kernel's row in the stats IS the launch roofline.avg_launch_ms times
# (the round-2 summary averaged it with the quarter-field launches of the side measurements)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
echo bench-trace-done
for model in smooth quarter zero invalid; do
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${model}_$n -- python3 $R/tools/prof_fi.py $model 196 > /dev/null 2>&1 || exit 1
  done
  echo pmc-$model-done
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_smooth_SQ -- python3 $R/tools/prof_fi.py smooth 196 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_FETCH_SIZE -- /tmp/fetchcal > /dev/null 2>&1 || exit 1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/cal_RDREQ -- /tmp/fetchcal > /dev/null 2>&1 || exit 1
echo calibration-done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proj -- python3 $R/tools/bench_proj.py --flows smooth,quarter --iters 50 > $OUT/proj_bench.log 2>&1 || exit 1
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/gate_proj_$n -- python3 $R/tools/bench_proj.py --flows smooth --iters 20 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/gate_fi3_$n -- python3 $R/tools/prof_fi.py smooth 3 > /dev/null 2>&1 || exit 1
done
echo gate-pmc-done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/projbatch -- python3 $R/tools/bench_r4.py --what projbatch --flows smooth --iters 30 > $OUT/projbatch_bench.log 2>&1 || exit 1
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  n=$(echo $ctr | cut -d' ' -f1)
  timeout -k 5 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/batch_proj_$n -- python3 $R/tools/bench_r4.py --what projbatch --flows smooth --iters 6 > /dev/null 2>&1 || exit 1
done
echo projbatch-done
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  n=$(echo $ctr | cut -d' ' -f1)
  timeout -k 5 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/multi3_$n -- python3 $R/tools/prof_multi.py smooth 3 > /dev/null 2>&1 || exit 1
done
echo multi-pmc-done
echo collected > $OUT/done
