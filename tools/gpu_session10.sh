#!/bin/bash
mkdir -p gpurun_out/s10
timeout 300 python tools/diag/gemm_bench.py > gpurun_out/s10/base.log 2>&1; cat gpurun_out/s10/base.log | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_train_encoder.py -m gpu -q --timeout 600 -x -k "pointwise_products" > gpurun_out/s10/pytest_mfma.log 2>&1; echo "mfma rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/s10/pytest_mfma.log | cut -c1-200 | head -20
