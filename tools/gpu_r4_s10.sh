# round-4 session 10: deferred split-K reduction of the pointwise weight gradients
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py -x -q 2>&1 | tail -8
TS_C4_ONLY=c4_phase2 timeout 600 python tools/bench_extra.py c4 2>&1 | tail -3 | cut -c1-1500
