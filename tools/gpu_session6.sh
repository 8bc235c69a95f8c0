#!/bin/bash
mkdir -p gpurun_out/s6
python -m pytest tests/test_gpu_tcs.py tests/test_gpu_citrinet.py -m gpu -q -x --timeout 900 > gpurun_out/s6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s6/pytest.log
tail -3 gpurun_out/s6/pytest.log
python tools/bench_tcs.py 2>&1 | grep -v amdgpu > gpurun_out/s6/bench_tcs.log; cat gpurun_out/s6/bench_tcs.log
