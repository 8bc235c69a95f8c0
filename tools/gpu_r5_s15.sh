cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_r2.py -x -q -m gpu -k dither 2>&1 | tail -30
