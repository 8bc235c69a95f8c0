#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s7
timeout 900 python -m pytest tests/test_gpu_r5.py tests/test_gpu_e2e.py tests/test_gpu_train.py -x -q > gpurun_out/r5s7/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5s7/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5s7/trace -- python3 bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline --no-trained-check > gpurun_out/r5s7/bench.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r5s7/trace/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r5s7/bench.log | head -2
