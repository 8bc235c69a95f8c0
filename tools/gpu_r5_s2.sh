#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s2
i=0
for cfg in "--lr 1e-3 --schedule 100:32:10:warm,2400:32:10:mix" "--lr 2e-3 --schedule 100:32:10:warm,1500:32:10:mix" "--lr 1e-3 --seed 1 --schedule 100:32:10:warm,2400:32:10:mix" "--lr 2e-3 --seed 1 --schedule 100:32:10:warm,1500:32:10:mix"; do
  i=$((i+1))
  timeout 600 python tools/train_margin_model.py --log-every 200 $cfg > gpurun_out/r5s2/train_$i.log 2>&1
  echo "== $cfg rc=$?"
  grep -v amdgpu.ids gpurun_out/r5s2/train_$i.log | cut -c1-3000 | tail -36
done
