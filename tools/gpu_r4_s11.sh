# round-4 session 11: conv0 with scalar signal operands
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_w2v.py tests/test_gpu_w2v_encoder.py tests/test_gpu_w2v_train.py -x -q 2>&1 | tail -5
timeout 600 python tools/bench_extra.py c5 2>&1 | tail -1 | cut -c1-900
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s11_c5 -- python3 tools/bench_extra.py c5 --no-check > gpurun_out/s11_c5.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/s11_c5/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
