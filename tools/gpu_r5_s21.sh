cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
{
for r in 1 2 3; do
  for v in old ""; do
    TS_LIB_VARIANT=$v timeout 300 python - <<'PY' 2>&1 | grep c3_ab
import os, sys; sys.path.insert(0, '.')
import torch, tools.bench_extra as be
r = be.c3(torch.device("cuda", 0), steps=20, check=False)
print('c3_ab', os.environ.get('TS_LIB_VARIANT') or 'new', round(r['ms_per_step'], 3), flush=True)
PY
  done
done
} | tee gpurun_out/r5b/c3_spill_ab.log
