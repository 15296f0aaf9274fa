#!/bin/bash
# A/B of projection builds under the kernel trace: tools/proj_ab.sh <tag> <flows> <libdir> [<libdir> ...]
# (<libdir> = a build directory inside the package: lib, lib_v<name> from tools/mkvariant.sh).  Through gpurun, repo root.
set -o pipefail
TAG=$1; FLOWS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG=$R/video-frame-interpolation-based-on-deformable-kernel-region_amd
mkdir -p $R/gpurun_out/projab_$TAG
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  OUT=$R/gpurun_out/projab_$TAG/$L
  mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/bench_proj.py --flows $FLOWS --iters 100 --lib $PKG/$L/libvfi_hip.so > $OUT/run.log 2>&1 || { echo "$L FAILED"; tail -5 $OUT/run.log; continue; }
  echo "== $L"
  grep -E '^proj|^dproj' $OUT/run.log
  python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].replace("void vfi::", "").replace("vfi::", "")[:34]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    if k.startswith("proj"):
        v2 = sorted(v)
        print("   %-36s n=%5d mean %7.2f median %7.2f min %7.2f" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
PY
done
