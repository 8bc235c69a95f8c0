#!/bin/bash
mkdir -p gpurun_out/s11
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s11/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s11/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s11/pytest.log | head -30
timeout 1200 python bench.py > gpurun_out/s11/bench.json 2> gpurun_out/s11/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/s11/bench.json") if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
    for k, v in d.get("extra", {}).items():
        if isinstance(v, dict) and "ms_per_step" in v: print(k, round(v["ms_per_step"], 3), "ms", round(v.get("value", 0), 1), v.get("unit"), v.get("check"), v.get("loss_first_last"))
        else: print(k, str(v)[:300])
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/s11/bench.err").read()[-2000:])
PY
bash tools/prof_round2.sh > gpurun_out/s11/prof.log 2>&1; tail -5 gpurun_out/s11/prof.log
