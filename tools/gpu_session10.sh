#!/bin/bash
mkdir -p gpurun_out/s10
for v in ""; do echo "== variant '$v'"; TS_LIB_VARIANT=$v python tools/diag/dw_bench.py 2>&1 | grep -v amdgpu; done
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_configs.py -m gpu -q --timeout 900 > gpurun_out/s10/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s10/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s10/pytest.log | head -30
timeout 600 python tools/bench_finetune.py --unfreeze --steps 20 --gemm-bf16 --graph 2>&1 | grep -v amdgpu | tail -2
