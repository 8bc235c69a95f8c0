#!/bin/bash
mkdir -p gpurun_out/s14
for i in 1 2 3; do
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/s14/pytest_$i.log 2>&1; echo "run $i rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s14/pytest_$i.log | head -5
done
