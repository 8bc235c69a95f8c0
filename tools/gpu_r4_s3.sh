#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tcs_chain.py tests/test_gpu_tcs.py tests/test_gpu_e2e.py -x -q > gpurun_out/s3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s3_tests.log
tail -8 gpurun_out/s3_tests.log
timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s3_chain.log 2>&1
cat gpurun_out/s3_chain.log | grep -v amdgpu.ids
TS_LIB_VARIANT=nonarrow timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s3_chain_nonarrow.log 2>&1
grep -v amdgpu.ids gpurun_out/s3_chain_nonarrow.log | head -12
