#!/bin/bash
mkdir -p gpurun_out/s9
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_r2.py tests/test_gpu_configs.py -m gpu -q --timeout 900 > gpurun_out/s9/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s9/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s9/pytest.log | head -30
timeout 600 python tools/bench_finetune.py --unfreeze --steps 20 --gemm-bf16 --graph 2>&1 | grep -v amdgpu | tail -2
bash tools/prof_ft2.sh --gemm-bf16 --graph > gpurun_out/s9/prof.log 2>&1; head -20 gpurun_out/s9/prof.log
