#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests/test_gpu_r2.py tests/test_gpu_configs.py tests/test_gpu_w2v_encoder.py -m gpu -q --timeout 900 > gpurun_out/s3/pytest_new.log 2>&1
echo "rc=$?" >> gpurun_out/s3/pytest_new.log
tail -40 gpurun_out/s3/pytest_new.log
