#!/bin/bash
mkdir -p gpurun_out/r3s7
timeout 1500 python bench.py --steps 50 --extra c4 --no-cpu-baseline > gpurun_out/r3s7/bench.json 2> gpurun_out/r3s7/bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r3s7/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3s7/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in d.get('extra',{}).items():
    if isinstance(v,dict):
        print(k, {kk:v[kk] for kk in v if kk in ('ms_per_step','value','error','n_collectives_per_step','exchange_ms_exposed','ms_per_step_without_exchange','loss_first_last','c_abi_calls_per_step')})
    else: print(k,v)
PY
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_encoder.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3s7/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3s7/pytest.log
