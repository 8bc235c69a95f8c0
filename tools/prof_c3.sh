# rocprofv3 kernel trace of the C3 (Citrinet-1024) bench:  bash tools/prof_c3.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_c3
rm -rf $O && mkdir -p $O
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_c3.py > $O/trace.log 2>&1
grep "^C3" $O/trace.log
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_c3/trace/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*int(r['TotalDurationNs'])/tot:5.1f}%")
PY
