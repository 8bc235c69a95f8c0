"""Same-box A/B of the C4 phase-2 step (local 32 x 10 s, bf16, one hipGraph): python tools/diag/c4_ab.py variant[:pw] ...   ('-' = product library;
:0 / :1 = ts_tcs_pointwise_select).  Each configuration runs in its own process, interleaved twice."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from thunder_speech_amd import _lib
pass
os.environ["TS_C4_ONLY"] = os.environ.get("TS_C4_ONLY", "c4_phase2")
from tools import bench_extra
r = bench_extra.c4(torch.device("cuda", 0), steps1=40, steps2=40)
print("RESULT", json.dumps({k: round(v["ms_per_step"], 3) for k, v in r.items()}))
''' % ROOT
def main(names):
    res = {n: [] for n in names}
    for rep in range(2):
        for n in names:
            var, _, pw = n.partition(":")
            env = dict(os.environ, TS_LIB_VARIANT="" if var == "-" else var, TS_PW_TILE=pw or "1")
            out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, cwd=ROOT)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            res[n].append(line[0][7:] if line else "FAILED " + out.stderr[-1500:])
    for n in names:
        print(f"{n:14s} " + "   ".join(res[n]), flush=True)
if __name__ == "__main__":
    main(sys.argv[1:] or ["-"])
