#!/bin/bash
mkdir -p gpurun_out
export TS_PW_SHORT=1
TS_PW_TILE=0 python tools/diag/pw_tile_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6c_pw_store.txt
TS_PW_TILE=1 python tools/diag/pw_tile_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6c_pw_store.txt
for v in st1 st2 st3 st4; do TS_LIB_VARIANT=$v TS_PW_TILE=1 python tools/diag/pw_tile_bench.py 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r6c_pw_store.txt
