"""Timing experiments on the split TCS kernel (diagnostic build -DTS_EXP, tools/variants.py exp=-DTS_EXP):
TS_EXP bits switch pieces of the kernel off (wrong results) to see what each piece costs on the C2 layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.bench_tcs import layer, bench

if __name__ == "__main__":
    B, T = 64, 751
    shapes = [("256->256 K33", (256, 256, 33, 0)), ("512->512 K63", (512, 512, 63, 0))]
    exps = [int(v) for v in os.environ.get("EXPS", "0,1,8,2,4,10,12,14,26,30,18").split(",")]
    for name, (ci, co, k, res) in shapes:
        L = layer(ci, co, k, res)
        for exp in exps:
            os.environ["TS_EXP"] = str(exp)
            bench(f"{name} exp={exp:2d}", L, B, T, iters=20)
