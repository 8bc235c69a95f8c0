#!/bin/bash
mkdir -p gpurun_out/r3s5
export TS_NO_V3=1
for v in "" d8 d16 d24 p3 d16p3; do
echo "== variant '$v'"
TS_LIB_VARIANT=$v timeout 300 python tools/bench_tcs.py 2>&1 | grep -E "512->512 K63  |256->256 K33  |512->512 K75  |estimated"
done
