#!/bin/bash
mkdir -p gpurun_out/s4
python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/s4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s4/pytest.log
tail -15 gpurun_out/s4/pytest.log
python bench.py > gpurun_out/s4/bench.json 2> gpurun_out/s4/bench.err; echo "bench rc=$?"
cat gpurun_out/s4/bench.json; tail -5 gpurun_out/s4/bench.err
