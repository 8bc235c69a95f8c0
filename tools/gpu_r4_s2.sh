#!/bin/bash
mkdir -p gpurun_out
./tools/diag/probe_stage > gpurun_out/s2_probe.log 2>&1; cat gpurun_out/s2_probe.log
python tools/diag/gemm_ceiling.py > gpurun_out/s2_gemm.log 2>&1; cat gpurun_out/s2_gemm.log
