#!/bin/bash
# deterministic training: two runs in one process each, checksums + transcript identity
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python tools/train_margin_model.py --log-every 1000 > gpurun_out/r6b_train$i.json 2> gpurun_out/r6b_train$i.err
  echo "run $i rc $?"; tail -2 gpurun_out/r6b_train$i.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r6b_train$i.json").read().splitlines() if l.startswith("{")][-1])
print(d["weights_sha256"], "flipped", d["frames_flipped"], "seq", d["collapsed_sequences_equal"], "strings", d["strings_equal"], "minmargin", d["min_fp32_margin"], "val", d["train"]["validation_margin"], "secs", d["train"]["seconds"])
PY
done
