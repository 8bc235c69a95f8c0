#!/bin/bash
mkdir -p gpurun_out/s10
timeout 300 python tools/diag/gemm_bench.py 2>&1 | grep -v amdgpu | grep "tcs\|Trace\|Error"
