#!/bin/bash
mkdir -p gpurun_out/r3s2
for v in exp exp4; do
TS_LIB_VARIANT=$v timeout 900 python tools/diag/exp_split.py > gpurun_out/r3s2/$v.log 2>&1; echo "$v rc=$?"
done
