#!/bin/bash
mkdir -p gpurun_out/r3s1
TS_LIB_VARIANT=exp timeout 900 python tools/diag/exp_split.py > gpurun_out/r3s1/exp.log 2>&1; echo "exp rc=$?"
timeout 600 python tools/bench_tcs.py > gpurun_out/r3s1/bench_tcs.log 2>&1; echo "bench_tcs rc=$?"
tail -4 gpurun_out/r3s1/bench_tcs.log
