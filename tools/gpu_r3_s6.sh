#!/bin/bash
mkdir -p gpurun_out/r3s6
timeout 1200 python -m pytest tests/test_gpu_nemo_e2e.py tests/test_gpu_train_encoder.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "nemo or frozen_schedule or refused or c2_full" > gpurun_out/r3s6/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r3s6/pytest.log


