#!/bin/bash
set -x
timeout 900 python -m pytest tests/test_gpu_tcs.py tests/test_gpu_citrinet.py tests/test_gpu_e2e.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -5
timeout 600 bash tools/diag/c3_ab.sh 2>&1 | tail -8
timeout 300 python tools/ab_encoder.py pw_wide=1 pw_wide=0 2>&1 | tail -6
