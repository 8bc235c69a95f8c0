#!/bin/bash
mkdir -p gpurun_out/r3s9
timeout 1200 python -m pytest tests/test_gpu_w2v.py tests/test_gpu_w2v_encoder.py tests/test_huggingface_loader.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3s9/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3s9/pytest.log
timeout 600 python tools/bench_c5.py --check > gpurun_out/r3s9/c5_ours.log 2>&1; tail -3 gpurun_out/r3s9/c5_ours.log
TS_W2V_VENDOR_GEMM=1 timeout 600 python tools/bench_c5.py > gpurun_out/r3s9/c5_vendor.log 2>&1; tail -2 gpurun_out/r3s9/c5_vendor.log
