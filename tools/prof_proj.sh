#!/bin/bash
# kernel trace of the projection bench: tools/prof_proj.sh <tag> [flows]   (through gpurun, from the repo root)
set -o pipefail
TAG=${1:-x}; FLOWS=${2:-smooth}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/projtrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_proj.py --flows $FLOWS --iters 50 $3 $4 > $OUT/run.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-62s n=%5d mean %8.2f us  median %8.2f  min %8.2f" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
PY
cat $OUT/run.log
