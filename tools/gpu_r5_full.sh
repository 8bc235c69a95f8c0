#!/bin/bash
# full GPU suite + the driver's bench command, as the round-end run does
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5full
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5full/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5full/smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5full/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5full/pytest.log
timeout 1200 python bench.py > gpurun_out/r5full/bench.json 2> gpurun_out/r5full/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
lines=[l for l in open('gpurun_out/r5full/bench.json').read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_world_size')}, 'frac', d['roofline']['frac'], 'enc', d['roofline']['encoder_ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('check', {k:v for k,v in d['check'].items() if k!='vs'})
ct=d.get('check_trained',{}); print('check_trained', {k:v for k,v in ct.items() if k not in('vs','flipped_frames','example')})
for k,v in d.get('extra',{}).items():
    if isinstance(v, dict): print(k, v.get('ms_per_step'), v.get('error'), v.get('projected_speedup_8'), v.get('local32_ms_per_step'))
PY
