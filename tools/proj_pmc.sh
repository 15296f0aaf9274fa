#!/bin/bash
# PMC passes over the projection kernels of one build: tools/proj_pmc.sh <tag> <libdir> [flow]   (through gpurun, repo root)
# Counters go in separate passes with --kernel-trace only (gpurun refuses --pmc beside the API traces).
set -o pipefail
TAG=$1; L=$2; FLOW=${3:-smooth}
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/video-frame-interpolation-based-on-deformable-kernel-region_amd
OUT=$R/gpurun_out/projpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_proj.py --flows $FLOW --iters 20 --lib $PKG/$L/libvfi_hip.so > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("void vfi::", "").replace("vfi::", "")[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(agg.items()):
    if not k.startswith("proj"): continue
    print(k)
    out[k] = {}
    for c, v in sorted(d.items()):
        out[k][c] = sum(v) / len(v)
        print("    %-26s n=%4d mean %14.1f" % (c, len(v), sum(v) / len(v)))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
