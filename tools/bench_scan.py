import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_tcs import layer, bench
for B in (42,):
    for cin in (64, 128, 256, 512, 1024):
        bench(f"B{B} {cin}->256 K33", layer(cin, 256, 33, 0), B, 751)
    for cin in (64, 256, 512):
        bench(f"B{B} {cin}->256 K1 pointwise", layer(cin, 256, 1, 0, separable=False), B, 751)
    for k in (5, 33, 75):
        bench(f"B{B} 256->256 K{k}", layer(256, 256, k, 0), B, 751)
    for cin in (64, 256, 512):
        bench(f"B21 {cin}->512 K33 (wide)", layer(cin, 512, 33, 0), 21, 751)
