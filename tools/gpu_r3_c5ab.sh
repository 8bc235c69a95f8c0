# C5 A/B in one session: own GEMM with packed weights / own GEMM through LDS / vendor GEMMs, twice each, interleaved
for i in 1 2; do
  for v in packed lds vendor; do
    unset TS_W2V_NO_FRAG TS_W2V_VENDOR_GEMM
    [ $v = lds ] && export TS_W2V_NO_FRAG=1
    [ $v = vendor ] && export TS_W2V_VENDOR_GEMM=1
    timeout 300 python tools/bench_extra.py c5 --no-check 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)['c5']; print('$v', round(r['ms_per_step'], 3), 'ms/step')
"
  done
done
