#!/bin/bash
mkdir -p gpurun_out/r3s10
timeout 1500 python tools/bench_extra.py c4 c5 > gpurun_out/r3s10/extra.json 2> gpurun_out/r3s10/extra.err; echo "rc=$?"; tail -3 gpurun_out/r3s10/extra.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3s10/extra.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict):
        print(k, {kk:v[kk] for kk in v if kk in ('ms_per_step','value','error','loss_first_last','c_abi_calls_per_step','check')})
PY
