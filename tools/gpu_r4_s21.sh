cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | grep -E "GROUP|Error|error" | head
import os, sys, torch
sys.path.insert(0, "tools")
os.environ["TS_C4_ONLY"] = "c4_phase2"
import bench_extra
from thunder_speech_amd import train_ops
for grp in (True, False, True, False):
    train_ops.GROUP_WGRAD = grp
    r = bench_extra.c4(torch.device("cuda", 0))
    print("GROUP_WGRAD", grp, round(r["c4_phase2"]["ms_per_step"], 3))
PY
