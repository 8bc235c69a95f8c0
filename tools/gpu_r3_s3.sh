#!/bin/bash
mkdir -p gpurun_out/r3s3
EXPS=0,30,32 TS_LIB_VARIANT=exp timeout 900 python tools/diag/exp_split.py > gpurun_out/r3s3/exp.log 2>&1; echo "rc=$?"
cat gpurun_out/r3s3/exp.log
