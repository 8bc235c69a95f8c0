#!/bin/bash
mkdir -p gpurun_out/s13
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s13/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s13/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/s13/pytest.log | head
for i in 1 2; do timeout 600 python tools/bench_finetune.py --unfreeze --steps 30 --gemm-bf16 --graph 2>&1 | grep "^C4"; done
timeout 600 python -c "
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -1
