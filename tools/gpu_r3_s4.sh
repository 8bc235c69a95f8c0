#!/bin/bash
mkdir -p gpurun_out/r3s4
EXPS=0,1,8,2,4,6,10,12,14,30 TS_LIB_VARIANT=exp timeout 900 python tools/diag/exp_split.py > gpurun_out/r3s4/exp.log 2>&1; echo "rc=$?"
cat gpurun_out/r3s4/exp.log
