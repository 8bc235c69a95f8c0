# round-4 session 9: segmented training graphs + c4_ddp with loop-back exchange
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_encoder.py -x -q -k "graphed" 2>&1 | tail -15
timeout 900 python - <<'PY' 2>&1 | tail -30
import json, sys, torch
sys.path.insert(0, "tools")
import bench_extra
r = bench_extra.c4_ddp(torch.device("cuda", 0))
print(json.dumps(r, indent=1))
PY
