#!/bin/bash
# round 4, session 1: chain kernel parity + A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tcs_chain.py tests/test_gpu_tcs.py -x -q > gpurun_out/s1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s1_tests.log
tail -15 gpurun_out/s1_tests.log
timeout 600 python tools/bench_chain.py --steps 30 > gpurun_out/s1_chain.log 2>&1
tail -40 gpurun_out/s1_chain.log
