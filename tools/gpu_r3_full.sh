#!/bin/bash
mkdir -p gpurun_out/r3full
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > gpurun_out/r3full/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r3full/pytest.log
