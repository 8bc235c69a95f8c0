#!/bin/bash
# round 6, first GPU call: child-process probe, the new tests, the default bench line
mkdir -p gpurun_out
python - > gpurun_out/r6a_probe.log 2>&1 <<'PY'
import subprocess, sys, torch
torch.zeros(1, device="cuda").add_(1); torch.cuda.synchronize()
print("gpu initialised; starting a child process")
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True)
print("rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
PY
cat gpurun_out/r6a_probe.log
timeout 1500 python -m pytest tests/test_gpu_r6.py -x -q > gpurun_out/r6a_tests.log 2>&1
tail -30 gpurun_out/r6a_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err
echo "bench rc $?"; tail -5 gpurun_out/r6a_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6a_bench.json").read().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["encoder_ms"])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "processes", "threads", "host_cores", "sweep_wall_s")}, [ (p.get("processes"), p.get("value")) for p in d["cpu_baseline"]["sweep"]])
print("predict_api", json.dumps(d.get("predict_api"))[:1500])
print("check", d.get("check", {}).get("collapsed_sequences_equal"), d.get("check_trained", {}).get("strings_equal"))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict): print(k, v.get("ms_per_step"), v.get("error"), v.get("projected_speedup_8"))
print("scaling", d.get("scaling_block"))
PY
