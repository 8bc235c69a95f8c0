#!/bin/bash
mkdir -p gpurun_out/s5
python -m pytest tests/test_gpu_tcs.py tests/test_gpu_e2e.py tests/test_gpu_citrinet.py -m gpu -q -x --timeout 900 > gpurun_out/s5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s5/pytest.log
tail -5 gpurun_out/s5/pytest.log
python tools/bench_tcs.py > gpurun_out/s5/bench_tcs.log 2>&1; cat gpurun_out/s5/bench_tcs.log | grep -v amdgpu
TS_LIB_VARIANT=stamp python tools/diag/stamp_dump.py 512 512 63 2>&1 | grep -v amdgpu | head -14 > gpurun_out/s5/stamps.log; cat gpurun_out/s5/stamps.log
