#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s10
timeout 900 python -m pytest tests/test_gpu_frontend_decode_ctc.py tests/test_gpu_reference_sweeps.py tests/test_gpu_r2.py -x -q > gpurun_out/r5s10/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5s10/pytest.log
for v in "" oldfe "" oldfe; do TS_LIB_VARIANT=$v timeout 300 python tools/diag/fe_time.py 2>&1 | tail -1; done
