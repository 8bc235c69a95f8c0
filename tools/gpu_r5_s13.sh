cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
timeout 900 python tools/diag/gemm_bk_ab.py > gpurun_out/r5b/gemm_switchoff.log 2>&1; echo "rc=$?"; tail -7 gpurun_out/r5b/gemm_switchoff.log
timeout 600 python tools/diag/fe_time.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_frontend_decode_ctc.py tests/test_gpu_r2.py tests/test_gpu_reference_sweeps.py -x -q -m gpu 2>&1 | tail -5
