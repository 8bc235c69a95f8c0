# round-4 session 12: squeeze-excite tail in the residual launch's epilogue; conv0 packed GELU + parallel finalize
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_citrinet.py tests/test_gpu_configs.py tests/test_gpu_w2v.py tests/test_gpu_w2v_encoder.py tests/test_capi_host.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -k "c3 or C3 or citrinet or c5 or w2v" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tail -8
import json, sys, torch
sys.path.insert(0, "tools")
import bench_extra
import thunder_speech_amd.citrinet.blocks as cb
dev = torch.device("cuda", 0)
for fuse in (True, False, True, False):
    cb.FUSE_SE_TAIL = fuse
    r = bench_extra.c3(dev, check=fuse)
    print("FUSE_SE_TAIL", fuse, "c3 ms/step", round(r["ms_per_step"], 3), r.get("check"))
r = bench_extra.c5(dev, check=True)
print("c5", round(r["ms_per_step"], 3), r.get("check"))
PY
