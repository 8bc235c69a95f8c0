#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s5
for cfg in "3 0.1" "2 0.1" "2 0.05" "2 0.2"; do
  set -- $cfg
  TS_C4_SEGMENTS=$1 TS_C4_FIRST_SHARE=$2 timeout 600 python tools/bench_extra.py c4_ddp > gpurun_out/r5s5/seg$1_$2.json 2> gpurun_out/r5s5/seg$1_$2.err
  python - "$1" "$2" <<'PY'
import json,sys
seg, share = sys.argv[1], sys.argv[2]
lines=[l for l in open(f'gpurun_out/r5s5/seg{seg}_{share}.json').read().splitlines() if l.startswith('{')]
if not lines:
    print('no json'); print(open(f'gpurun_out/r5s5/seg{seg}_{share}.err').read()[-1500:]); sys.exit(0)
d=json.loads(lines[-1])['c4_ddp']
if 'error' in d: print(d); sys.exit(0)
print(f'seg {seg} share {share}:', 't256_off %.2f t256 %.2f | t32_off %.3f t32_lb %.3f exposed %.3f | proj %.3f' % (d['ms_per_step_without_exchange'], d['ms_per_step'], d['local32_ms_per_step_without_exchange'], d['local32_ms_per_step'], d['local32_exchange_ms_exposed_loopback'], d['projected_speedup_8']), d['n_graphs'], d['n_buckets'], d['exposed_bucket_bytes_on_wire'])
PY
done
