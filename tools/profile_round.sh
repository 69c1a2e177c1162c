#!/bin/bash
# Refresh profiles/<round>/ on the GPU box:  bash tools/profile_round.sh r1
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> bench_n1_kernel_stats.csv (+ the bench line)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short bench run -> pmc_fetch_write_per_kernel.json
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
set -e
ROUND=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p $OUT $REPO/profiles/$ROUND
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_line.txt 2> $OUT/bench_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_${c}_err.txt
done
python3 - "$OUT" "$REPO/gpurun_out/profiles_$ROUND" <<'PY'
import csv, glob, json, os, sys, collections, shutil
out, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
shutil.copy(st[0], dst + "/bench_n1_kernel_stats.csv")
shutil.copy(out + "/bench_line.txt", dst + "/bench_n1_line.json")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for k, d in acc.items():
    name = k.split("(")[0]
    summ[name] = {f"{c}_KB_avg": sum(v) / len(v) for c, v in d.items()}
    summ[name]["launches"] = max(len(v) for v in d.values())
json.dump(summ, open(dst + "/pmc_fetch_write_per_kernel.json", "w"), indent=1, sort_keys=True)
print(open(out + "/bench_line.txt").read().strip()[:400])
PY
