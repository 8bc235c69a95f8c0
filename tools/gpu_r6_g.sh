#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TS_PROF_MARK=1 TS_C5FT_ONLY=bf16
O=gpurun_out/prof_c5ftb
rm -rf $O && mkdir -p $O
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/bench_extra.py c5_finetune > $O/log 2>&1
tail -2 $O/log | cut -c1-400
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/prof_c5ftb/t/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i in range(len(rows) - 1) if "counter_add" in rows[i]["Kernel_Name"] and "counter_add" in rows[i + 1]["Kernel_Name"]]
rows = [r for r in rows[marks[0] + 2: marks[-1]] if "counter_add" not in r["Kernel_Name"]]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r["Kernel_Name"][:90]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values()); span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
N = 12   # timed steps of the mixed-precision run (tools/bench_extra.py c5_finetune)
print("launches", len(rows) / N, "per step; kernel ms/step", tot / N / 1e6, "span ms/step", span / N / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{v[0] / N:7.1f} x {v[1] / v[0] / 1e3:8.1f} us = {v[1] / N / 1e6:7.2f} ms  {k}")
PY
