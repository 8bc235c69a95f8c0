#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s11
timeout 600 python bench.py --weights trained --no-extra --steps 50 > gpurun_out/r5s11/bench_trained.json 2> gpurun_out/r5s11/bench_trained.err; echo "bench trained rc=$?"
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/r5s11/bench_trained.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['config']['weights'], d['config']['trained'])
print('check', d['check']['collapsed_sequences_equal'], d['check']['argmax_equal_all_frames_frac'], d['check']['max_err_over_scale'])
print('check_trained', d['check_trained']['collapsed_sequences_equal'], d['check_trained']['frames_flipped'])
PY
timeout 600 python tools/bench_finetune.py --unfreeze --gemm-bf16 --graph 2>&1 | tail -2
timeout 600 python tools/bench_c5.py --check 2>&1 | tail -2
