#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s8
timeout 900 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_full_size.py -x -q -s -k "graphed or c4 or decoder or train" > gpurun_out/r5s8/pytest.log 2>&1; echo "pytest rc=$?"; grep "C4 local" gpurun_out/r5s8/pytest.log; tail -3 gpurun_out/r5s8/pytest.log
for i in 1 2; do
timeout 600 python tools/bench_extra.py c4_ddp 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['c4_ddp']
print('t256_off %.2f t256 %.2f | t32_off %.3f t32_lb %.3f exposed %.3f | proj %.3f' % (d['ms_per_step_without_exchange'], d['ms_per_step'], d['local32_ms_per_step_without_exchange'], d['local32_ms_per_step'], d['local32_exchange_ms_exposed_loopback'], d['projected_speedup_8']))"
done
