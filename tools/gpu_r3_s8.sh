#!/bin/bash
mkdir -p gpurun_out/r3s8
timeout 1500 python -m pytest tests/test_gpu_reference_sweeps.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3s8/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r3s8/pytest.log
