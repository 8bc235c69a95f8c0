#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s3
timeout 900 python -m pytest tests/test_gpu_trained_transcripts.py -x -q > gpurun_out/r5s3/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5s3/pytest.log
timeout 1500 python bench.py > gpurun_out/r5s3/bench.json 2> gpurun_out/r5s3/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5s3/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5s3/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_world_size')}, d['roofline']['frac'])
print('check', {k:v for k,v in d['check'].items() if k!='vs'})
ct=d.get('check_trained',{}); print('check_trained', {k:v for k,v in ct.items() if k not in('vs','flipped_frames','example')})
for k,v in d.get('extra',{}).items(): print(k, v.get('ms_per_step'), v.get('error'), v.get('projected_speedup_8'))
PY
