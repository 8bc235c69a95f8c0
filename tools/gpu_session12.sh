#!/bin/bash
mkdir -p gpurun_out/s12
timeout 900 python bench.py --no-extra > gpurun_out/s12/bench.json 2> gpurun_out/s12/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/s12/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["encoder_ms"], d["roofline"]["avg_launch_us"])
PY
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/s12/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s12/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s12/pytest.log | head
