cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_ft1
rm -rf $O && mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_finetune.py > $O/trace.log 2>&1
grep "^C4" $O/trace.log
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/prof_ft1/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
n = [int(r['Calls']) for r in rows if 'ctc_kernel' in r['Name']][0]
print("steps", n, "GPU ms/step", tot/1e6/n, "launches/step", sum(int(r['Calls']) for r in rows)/n)
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls'])/n:7.1f} {int(r['TotalDurationNs'])/1e6/n:8.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
