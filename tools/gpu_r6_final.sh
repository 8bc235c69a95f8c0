#!/bin/bash
# full validation of the tree: the -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f_tests.log 2>&1
echo "tests rc $?"; tail -4 gpurun_out/r6f_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6f_smoke.log 2>&1; tail -2 gpurun_out/r6f_smoke.log
timeout 900 python bench.py > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
echo "bench rc $?"; tail -3 gpurun_out/r6f_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6f_bench.json").read().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("encoder_ms"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "processes", "threads", "host_cores")})
print("predict_api", json.dumps(d.get("predict_api"))[:800])
print("check", d.get("check", {}).get("collapsed_sequences_equal"), d.get("check_trained", {}).get("strings_equal"))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict): print(k, v.get("ms_per_step"), v.get("error"), v.get("projected_speedup_8"))
PY
