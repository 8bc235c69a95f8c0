"""Same-box A/B of library variants on the C4 steps (tools/bench_extra.py c4): python tools/diag/c4_ab.py variantA variantB ... ('' = product); each in its
own process, interleaved twice."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
names = sys.argv[1:] or ["", "old"]
for rep in range(2):
    for n in names:
        out = subprocess.run([sys.executable, "tools/bench_extra.py", "c4", "--no-check"], env=dict(os.environ, TS_LIB_VARIANT=n), capture_output=True, text=True, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        d = json.loads(lines[-1]) if lines else {}
        print(f"{n or 'product':10s}", {k: round(v["ms_per_step"], 3) for k, v in d.items() if isinstance(v, dict) and "ms_per_step" in v} or out.stderr[-500:], flush=True)
