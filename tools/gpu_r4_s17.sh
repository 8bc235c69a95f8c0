cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/s17_bench.json 2> gpurun_out/s17_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/s17_bench.err
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/s17_bench.json") if l.startswith("{")][-1])
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["frac"], r["roofline"]["avg_launch_us"], r.get("check", {}).get("collapsed_sequences_equal"))
for k, v in r.get("extra", {}).items():
    if isinstance(v, dict): print(k, round(v.get("ms_per_step", 0), 3), v.get("projected_speedup_8"), v.get("error"))
PY
