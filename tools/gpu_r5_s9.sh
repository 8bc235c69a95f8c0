#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s9
for seed in 0 1 2 3; do
  timeout 600 python tools/train_margin_model.py --seed $seed --log-every 1000 > gpurun_out/r5s9/train_$seed.log 2>&1
  grep "after" gpurun_out/r5s9/train_$seed.log | tr '\n' ';'; echo
  python - $seed <<'PY'
import json,sys
l=[x for x in open(f'gpurun_out/r5s9/train_{sys.argv[1]}.log').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print({k:d[k] for k in ('collapsed_sequences_equal','strings_equal','frames_flipped','min_fp32_margin')}, d['label_error_rate_vs_ground_truth'], d['train']['steps'], round(d['train']['seconds'],1))
PY
done
