cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
timeout 600 python tools/diag/fe_time.py 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_frontend_decode_ctc.py tests/test_gpu_r2.py tests/test_gpu_reference_sweeps.py tests/test_gpu_train.py tests/test_gpu_train_encoder.py tests/test_gpu_e2e.py tests/test_gpu_gemm_nt.py -x -q -m gpu 2>&1 | tail -5
