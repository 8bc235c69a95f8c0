cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_tcs.py -x -q 2>&1 | tail -3
TS_C4_ONLY=c4_phase2 timeout 600 python tools/bench_extra.py c4 2>&1 | tail -1 | cut -c1-600
