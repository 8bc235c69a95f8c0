#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_frontend_decode_ctc.py tests/test_gpu_r2.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/diag/exp_fe.py 2>&1 | tail -2
TS_FE_WG_KERNEL=1 timeout 300 python tools/diag/exp_fe.py 2>&1 | tail -1
