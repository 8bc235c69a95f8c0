#!/bin/bash
mkdir -p gpurun_out/s2
export TS_LIB_VARIANT=stamp
for args in "512 1024 1" "512 512 63" "512 512 63 512" "256 256 33"; do
  echo "=== $args" >> gpurun_out/s2/stamps.log
  timeout 200 python tools/diag/stamp_dump.py $args >> gpurun_out/s2/stamps.log 2>&1
done
cat gpurun_out/s2/stamps.log
