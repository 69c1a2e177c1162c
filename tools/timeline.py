#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state fit step from a rocprofv3 --kernel-trace CSV (start, end, duration in
us relative to the end of the previous step's Adam; queue id shows which stream a kernel ran on).
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline
   python tools/timeline.py DIR"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_k")]
a0, a1 = adam[-3], adam[-2]
t0 = int(rows[a0]["End_Timestamp"])
for r in rows[a0 + 1:a1 + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:60]}")
