#!/bin/bash
mkdir -p gpurun_out/s12
for i in 1 2; do
for v in "" 1; do
TS_NO_LENGTHS_MAP=$v timeout 900 python bench.py --no-extra --no-cpu-baseline > gpurun_out/s12/bench_$v$i.json 2> gpurun_out/s12/bench.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/s12/bench_$v$i.json") if l.startswith("{")][-1])
print("nomap='$v'", round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 4), round(d["roofline"]["encoder_ms"], 4), round(d["roofline"]["avg_launch_us"], 2))
PY
done; done
