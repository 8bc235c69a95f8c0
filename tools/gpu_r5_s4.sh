#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s4
timeout 1500 python -m pytest tests/test_gpu_r5.py tests/test_gpu_train_encoder.py tests/test_gpu_citrinet.py tests/test_gpu_configs.py -x -q > gpurun_out/r5s4/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r5s4/pytest.log
timeout 300 python tools/bench_c3.py --steps 20 2>&1 | tail -1
TS_C4_ONLY=c4_phase2 timeout 600 python tools/bench_extra.py c4 2>&1 | tail -1 | cut -c1-600
