cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TS_PROF_MARK=1 TS_C4_ONLY=c4_phase2
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s16_c4 -- python3 tools/bench_extra.py c4 > gpurun_out/s16_c4.log 2>&1
tail -1 gpurun_out/s16_c4.log | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/s16_c4/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
MARK = "counter_add_kernel"
marks = [i for i in range(len(rows) - 1) if MARK in rows[i]["Kernel_Name"] and MARK in rows[i + 1]["Kernel_Name"]]
rows = [r for r in rows[marks[0] + 2: marks[-1]] if MARK not in r["Kernel_Name"]]
steps = sum(1 for r in rows if "ctc_kernel" in r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"][:64]; d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[k][0] += 1; agg[k][1] += d
tot = sum(v[1] for v in agg.values()); span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"steps {steps}  kernel ms/step {tot/steps/1e6:.3f}  wall ms/step {span/steps/1e6:.3f}  launches/step {len(rows)/steps:.0f}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:64s} n/step={c/steps:6.1f} avg={t/c/1e3:7.1f}us  ms/step={t/steps/1e6:6.3f}")
PY
