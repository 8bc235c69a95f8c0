# rocprofv3 kernel trace of the fine-tuning step (phase 2) in a given mode:  bash tools/prof_ft2.sh [--gemm-bf16]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_ft2
rm -rf $O && mkdir -p $O
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_finetune.py --unfreeze --steps 10 "$@" > $O/trace.log 2>&1
grep "^C4" $O/trace.log
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/prof_ft2/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
steps = 13
print(f"total kernel time per step {tot/1e6/steps:.2f} ms, launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/steps:7.1f} {int(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
os.system(f"cp {f} gpurun_out/prof_ft2/kernel_stats.csv")
PY
