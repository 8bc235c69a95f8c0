cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
timeout 900 python tools/diag/gemm_bk_ab.py > gpurun_out/r5b/gemm_bk_ab.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r5b/gemm_bk_ab.log
