#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s6
timeout 1500 python -m pytest tests/test_gpu_r5.py tests/test_gpu_tcs.py tests/test_gpu_e2e.py tests/test_gpu_citrinet.py tests/test_gpu_configs.py tests/test_gpu_nemo_e2e.py tests/test_capi_host.py tests/test_gpu_train.py -x -q > gpurun_out/r5s6/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5s6/pytest.log
timeout 300 python tools/enc_time.py 2>&1 | tail -1
timeout 600 python bench.py --no-extra --no-cpu-baseline --no-trained-check --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['encoder_ms'], d['roofline']['frac'])"
