#!/bin/bash
mkdir -p gpurun_out/s10
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s10/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s10/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s10/pytest.log | head
for v in "" 1; do
  if [ -n "$v" ]; then export TS_PW_SPLIT=1; else unset TS_PW_SPLIT; fi
  echo "== TS_PW_SPLIT='$v'"
  python tools/bench_c3.py 2>&1 | grep "^C3"
done
