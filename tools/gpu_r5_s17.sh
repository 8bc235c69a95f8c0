cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_w2v_encoder.py tests/test_huggingface_loader.py -q -m gpu -k "d2v or families" 2>&1 | grep -E "^E  |Error|assert|FAILED|passed|failed" | cut -c1-260 | head -60
