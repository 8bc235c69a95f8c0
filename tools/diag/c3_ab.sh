#!/bin/bash
# Same-box A/B of the C3 (Citrinet-1024) encoder step: wide-frame pointwise tiles on / off.
for i in 1 2; do for w in 1 0; do python - <<PY
import json, subprocess, sys, os
sys.path.insert(0, ".")
sys.argv = ["bench_extra.py", "c3", "--no-check"]
from thunder_speech_amd import _lib
_lib.lib().ts_tcs_pointwise_wide($w)
import io, contextlib, runpy
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("tools/bench_extra.py", run_name="__main__")
    except SystemExit:
        pass
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        d = json.loads(line)
        print("wide=$w", round(d["c3"]["ms_per_step"], 3))
PY
done; done
