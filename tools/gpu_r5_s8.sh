#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s8
timeout 1200 python -m pytest tests/test_gpu_r5.py tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_full_size.py tests/test_gpu_r2.py -x -q > gpurun_out/r5s8/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5s8/pytest.log
for i in 1 2; do
timeout 600 python tools/bench_extra.py c4_ddp 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['c4_ddp']
print('t256_off %.2f t256 %.2f | t32_off %.3f t32_lb %.3f exposed %.3f | proj %.3f' % (d['ms_per_step_without_exchange'], d['ms_per_step'], d['local32_ms_per_step_without_exchange'], d['local32_ms_per_step'], d['local32_exchange_ms_exposed_loopback'], d['projected_speedup_8']))"
done
TS_C4_ONLY=c4_phase2 timeout 600 python tools/bench_extra.py c4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
