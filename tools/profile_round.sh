#!/bin/bash
# Refresh profiles/<round>/ on the GPU box:  bash tools/profile_round.sh r2
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> bench_n1_kernel_stats.csv (+ the bench line)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short bench run -> pmc_fetch_write_per_kernel.json
#   3. the log-mel front end (not part of the timed fit step): kernel stats, FETCH/WRITE and SQ counters of
#      tools/logmel_bench.py -> logmel_kernel_stats.csv, logmel_pmc.json, logmel_line.txt
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
set -e
ROUND=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$ROUND
DST=$REPO/gpurun_out/profiles_$ROUND
mkdir -p $OUT $DST
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_line.txt 2> $OUT/bench_err.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_${c}_err.txt
done
# 4. the inference path: eval-mode forward of config 2 (tools/eval_bench.py) -> eval_kernel_stats.csv, eval_line.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval_stats -- python3 $REPO/tools/eval_bench.py --reps 30 > $OUT/eval_line.txt 2> $OUT/eval_err.txt
# 5. BASELINE config 5 at its full per-GPU size (tools/cfg_timeline.py 5): kernel stats + the timeline of one step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -- python3 $REPO/tools/cfg_timeline.py 5 > /dev/null 2> $OUT/cfg5_err.txt
python3 $REPO/tools/timeline.py $OUT/cfg5 > $OUT/cfg5_timeline.txt
# 6. the timeline of one config-2 step (critical-path table)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/tl_err.txt
python3 $REPO/tools/timeline.py $OUT/tl > $OUT/step_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_stats -- python3 $REPO/tools/logmel_bench.py --reps 20 > $OUT/logmel_line.txt 2> $OUT/lm_err.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  d=$OUT/lm_pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $REPO/tools/logmel_bench.py --reps 3 > /dev/null 2> $d.err
done
python3 - "$OUT" "$DST" <<'PY'
import csv, glob, json, os, sys, collections, shutil
out, dst = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
shutil.copy(st[0], dst + "/bench_n1_kernel_stats.csv")
shutil.copy(out + "/bench_line.txt", dst + "/bench_n1_line.json")
def summarise(pattern, big_grid_only=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if big_grid_only and (big_grid_only not in r["Kernel_Name"] or int(r["Grid_Size"]) < 100000):
                continue
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = {}
    for k, d in acc.items():
        name = k.replace("(anonymous namespace)::", "").split("(")[0]
        summ[name] = {c: sum(v) / len(v) for c, v in d.items()}
        summ[name]["launches"] = max(len(v) for v in d.values())
    return summ
s = summarise(out + "/pmc_*/**/*counter_collection.csv")
json.dump({k: {(c + "_KB_avg" if c.endswith("_SIZE") else c): v for c, v in d.items()} for k, d in s.items()},
          open(dst + "/pmc_fetch_write_per_kernel.json", "w"), indent=1, sort_keys=True)
for sub, name in (("eval_stats", "eval_kernel_stats.csv"), ("cfg5", "cfg5_kernel_stats.csv")):
    g = glob.glob(out + "/" + sub + "/**/*kernel_stats.csv", recursive=True)
    if g:
        shutil.copy(g[0], dst + "/" + name)
for f in ("eval_line.txt", "cfg5_timeline.txt", "step_timeline.txt"):
    if os.path.exists(out + "/" + f):
        shutil.copy(out + "/" + f, dst + "/" + f)
lm = glob.glob(out + "/lm_stats/**/*kernel_stats.csv", recursive=True)
if lm:
    shutil.copy(lm[0], dst + "/logmel_kernel_stats.csv")
    shutil.copy(out + "/logmel_line.txt", dst + "/logmel_line.txt")
    json.dump(summarise(out + "/lm_pmc_*/**/*counter_collection.csv", "logmel_fft_k"), open(dst + "/logmel_pmc.json", "w"), indent=1, sort_keys=True)
print(open(out + "/bench_line.txt").read().strip()[:400])
print(open(out + "/logmel_line.txt").read().strip().splitlines()[-1])
PY
