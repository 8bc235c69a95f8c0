# round-4 session 13: SE tail test + A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_citrinet.py -x -q 2>&1 | tail -12
timeout 600 python - <<'PY' 2>&1 | tail -8
import json, sys, torch
sys.path.insert(0, "tools")
import bench_extra
import thunder_speech_amd.citrinet.blocks as cb
dev = torch.device("cuda", 0)
for fuse in (True, False, True, False):
    cb.FUSE_SE_TAIL = fuse
    r = bench_extra.c3(dev, check=fuse)
    print("FUSE_SE_TAIL", fuse, "c3 ms/step", round(r["ms_per_step"], 3), (r.get("check") or {}).get("max_err_over_scale"))
PY
