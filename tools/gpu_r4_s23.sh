cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | grep -E "TILE|Error|error" | head
import os, sys, torch
sys.path.insert(0, "tools")
os.environ["TS_C4_ONLY"] = "c4_phase2"
import bench_extra
from thunder_speech_amd import train_ops
for mode in (True, False, True, False):
    train_ops.TILE_STATS = mode
    r = bench_extra.c4(torch.device("cuda", 0), local_batch=256, steps1=4, steps2=8)
    print("TILE_STATS", mode, "local 256:", round(r["c4_phase2"]["ms_per_step"], 3))
PY
