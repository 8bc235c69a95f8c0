#!/bin/bash
# round-2 session 1: baseline state on a fresh box + cheap split-kernel variants
mkdir -p gpurun_out/s1
python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
for v in "" ring4 pprio cprio pcprio; do
  echo "== variant '$v'" >> gpurun_out/s1/tcs_variants.log
  TS_LIB_VARIANT=$v timeout 300 python tools/bench_tcs.py >> gpurun_out/s1/tcs_variants.log 2>&1
done
tail -3 gpurun_out/s1/pytest.log; cat gpurun_out/s1/bench.json; grep -E "variant|estimated" gpurun_out/s1/tcs_variants.log
