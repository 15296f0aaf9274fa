#!/bin/bash
# LDS bank-conflict counters of the staged FilterInterpolation kernels on controlled flow fields, for one or more builds:
#   tools/lds_conflicts.sh <f16|f32> <libdir> [<libdir> ...]     (repo root, through gpurun; libdir = lib, lib_v<name>)
WHAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/video-frame-interpolation-based-on-deformable-kernel-region_amd
O=$R/gpurun_out/ldsconf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  for m in zero rowstep smooth quarter; do
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/${WHAT}_${L}_$m -- python3 $R/tools/prof_misc.py $WHAT $m --lib $PKG/$L/libvfi_hip.so > $O/${WHAT}_${L}_$m.log 2>&1 || { tail -3 $O/${WHAT}_${L}_$m.log; exit 1; }
  done
done
python3 - $O $WHAT "$@" <<'PY'
import csv, glob, sys, collections
o, what = sys.argv[1], sys.argv[2]
for L in sys.argv[3:]:
    for m in ("zero", "rowstep", "smooth", "quarter"):
        a = collections.defaultdict(list); dur = []
        for f in glob.glob("%s/%s_%s_%s/**/*counter_collection.csv" % (o, what, L, m), recursive=True):
            for r in csv.DictReader(open(f)):
                if "fi_forward_ori_lds" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob("%s/%s_%s_%s/**/*kernel_trace.csv" % (o, what, L, m), recursive=True):
            for r in csv.DictReader(open(f)):
                if "fi_forward_ori_lds" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        v = {k: sum(x) / len(x) / 1e6 for k, x in a.items()}
        print("%-4s %-12s %-8s conflict %7.1f M  active %7.1f M  insts %6.1f M   %7.1f us" % (what, L, m, v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_INSTS_LDS", 0), sum(dur) / max(1, len(dur))))
PY
