#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tcs_chain.py tests/test_gpu_tcs.py tests/test_gpu_e2e.py tests/test_gpu_citrinet.py tests/test_gpu_configs.py -x -q > gpurun_out/s4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s4_tests.log
tail -12 gpurun_out/s4_tests.log
timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s4_chain.log 2>&1
grep -v amdgpu.ids gpurun_out/s4_chain.log
TS_LIB_VARIANT=noid2 timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s4_chain_noid2.log 2>&1
grep -v amdgpu.ids gpurun_out/s4_chain_noid2.log
