"""Same-box A/B of the C4 steps with and without deterministic gradients (train_ops.set_deterministic: ordered partials + one reduce launch per
depthwise layer instead of float atomics); each mode in its own process, interleaved twice."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, json, io, contextlib, runpy, torch
sys.path.insert(0, %r)
from thunder_speech_amd import train_ops
if sys.argv[1] == "1":
    train_ops.set_deterministic(True, "cuda:0")
sys.argv = ["bench_extra.py", "c4", "--no-check"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("tools/bench_extra.py", run_name="__main__")
    except SystemExit:
        pass
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        d = json.loads(line)
        print("RESULT", json.dumps({k: round(v["ms_per_step"], 3) for k, v in d.items() if isinstance(v, dict) and "ms_per_step" in v}))
''' % ROOT
for rep in range(2):
    for mode in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", CODE, mode], capture_output=True, text=True, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print("deterministic" if mode == "1" else "atomics      ", lines[0][7:] if lines else out.stderr[-800:], flush=True)
