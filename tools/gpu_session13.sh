#!/bin/bash
mkdir -p gpurun_out/s13
timeout 900 python -m pytest tests/test_gpu_train_encoder.py -m gpu -q --timeout 600 -x -k "depthwise_pair" > gpurun_out/s13/pytest_dw.log 2>&1; echo "dw rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/s13/pytest_dw.log | cut -c1-200 | head -12
python tools/diag/dw_bench.py 2>&1 | grep -v amdgpu
TS_DW_NO_MFMA=1 python tools/diag/dw_bench.py 2>&1 | grep -v amdgpu | grep bwd
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_r2.py -m gpu -q --timeout 900 > gpurun_out/s13/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s13/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s13/pytest.log | head
for i in 1 2; do timeout 600 python tools/bench_finetune.py --unfreeze --steps 30 --gemm-bf16 --graph 2>&1 | grep "^C4"; done
