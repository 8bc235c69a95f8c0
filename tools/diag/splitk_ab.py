"""Sweep of huggingface/train.py SPLITK_TARGET_WGS on the mixed-precision fine-tuning step: python tools/diag/splitk_ab.py <target>."""
import os, sys, json, io, contextlib, runpy
os.chdir(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.getcwd())
from thunder_speech_amd.huggingface import train as T
T.SPLITK_TARGET_WGS = int(sys.argv[1])
os.environ["TS_C5FT_ONLY"] = "bf16"
sys.argv = ["bench_extra.py", "c5_finetune"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try: runpy.run_path("tools/bench_extra.py", run_name="__main__")
    except SystemExit: pass
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        print(T.SPLITK_TARGET_WGS, json.loads(line)["c5_finetune_bf16"]["ms_per_step"])
