cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5b
{
for r in 1 2 3; do
  TS_LIB_VARIANT=old timeout 300 python tools/enc_time.py 2>&1 | grep "encoder ms"
  timeout 300 python tools/enc_time.py 2>&1 | grep "encoder ms"
done
} | tee gpurun_out/r5b/spill_ab.log
timeout 1500 python -m pytest tests/test_gpu_tcs.py tests/test_gpu_e2e.py tests/test_gpu_citrinet.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
