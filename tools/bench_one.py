"""Run a few launches of one TCS layer shape (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_tcs import layer, bench
cin, cout, k, res = [int(v) for v in sys.argv[1:5]]
bench(f"{cin}->{cout} K{k} res{res}", layer(cin, cout, k, res, separable=k > 1), 64, 751, iters=int(sys.argv[5]) if len(sys.argv) > 5 else 5)
