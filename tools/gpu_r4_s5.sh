#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tcs_chain.py tests/test_gpu_tcs.py tests/test_gpu_e2e.py -x -q > gpurun_out/s5_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s5_tests.log
tail -4 gpurun_out/s5_tests.log
timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s5_chain.log 2>&1
grep -v amdgpu.ids gpurun_out/s5_chain.log
