#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s12
mkdir -p $O
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline > $O/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/s12/bench/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
