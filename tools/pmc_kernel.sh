#!/bin/bash
# Per-kernel PMC counters for one command:  bash tools/pmc_kernel.sh <tag> "<counter list>" -- python3 script.py args
# One --pmc pass with --kernel-trace only (no other tracing domains, as the pool requires); prints per-kernel averages.
set -e
TAG=$1; CTRS=$2; shift 3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt || { tail -5 $OUT/stderr.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    n = max(len(v) for v in d.values())
    print(f"{k}  (x{n})")
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}")
PY
