cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train_encoder.py tests/test_gpu_train.py tests/test_gpu_tcs.py tests/test_capi_host.py -x -q 2>&1 | tail -4
timeout 900 python - <<'PY' 2>&1 | grep -E "TILE|Error|error" | head
import os, sys, torch
sys.path.insert(0, "tools")
os.environ["TS_C4_ONLY"] = "c4_phase2"
import bench_extra
from thunder_speech_amd import train_ops
for mode in (True, False, True, False):
    train_ops.TILE_STATS = mode
    r = bench_extra.c4(torch.device("cuda", 0))
    print("TILE_STATS", mode, round(r["c4_phase2"]["ms_per_step"], 3), r["c4_phase2"]["loss_first_last"])
PY
