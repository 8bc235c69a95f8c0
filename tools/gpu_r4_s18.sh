cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | tail -14
import json, sys, torch
sys.path.insert(0, "tools")
import bench_extra
for fs in (0.1, None):
    r = bench_extra.c4_ddp(torch.device("cuda", 0), first_share=fs)
    print("first_share", fs, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("ms_per_step", "ms_per_step_without_exchange", "exchange_ms_exposed", "local32_ms_per_step", "local32_ms_per_step_without_exchange", "local32_exchange_ms_exposed_loopback", "exposed_bucket_bytes_on_wire", "projected_speedup_8")})
PY
