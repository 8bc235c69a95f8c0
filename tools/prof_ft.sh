# rocprofv3 kernel trace of the fine-tuning step (phase 2, encoder unfrozen):  bash tools/prof_ft.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_ft
rm -rf $O && mkdir -p $O
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_finetune.py --unfreeze > $O/trace.log 2>&1
grep "^C4" $O/trace.log
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/prof_ft/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*int(r['TotalDurationNs'])/tot:5.1f}%")
PY
