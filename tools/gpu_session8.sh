#!/bin/bash
mkdir -p gpurun_out/s8
python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_r2.py tests/test_gpu_configs.py -m gpu -q --timeout 900 > gpurun_out/s8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s8/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s8/pytest.log | head -30
python tools/bench_finetune.py --unfreeze --steps 10 2>&1 | grep -v amdgpu | tail -2
python tools/bench_finetune.py --unfreeze --steps 10 --gemm-bf16 2>&1 | grep -v amdgpu | tail -2
