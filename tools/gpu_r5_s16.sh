cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
timeout 600 python tools/diag/c3_shapes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b/c3_shapes.log
