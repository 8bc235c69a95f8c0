"""Timed-region markers for profiled runs: with TS_PROF_MARK=1 every timed loop of bench.py / tools/bench_extra.py is bracketed by TWO
back-to-back launches of a tiny library kernel (ts::counter_add_kernel via ts_counter_add; a single launch of it is also the
replay-nonce bump at the head of a graphed training step, a pair is only ever a marker), so that tools/prof_round_summary.py can
drop warm-up, graph-capture and one-time packing dispatches from the kernel trace and report per-step shares of the timed region only."""
import os

_CELL = {}


def mark(device="cuda"):
    if not os.environ.get("TS_PROF_MARK"):
        return
    import torch
    from thunder_speech_amd import _lib
    dev = torch.device(device)
    key = str(dev)
    if key not in _CELL:
        _CELL[key] = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    for _ in range(2):
        _lib.check(_lib.lib().ts_counter_add(_CELL[key].data_ptr(), 1, torch.cuda.current_stream(dev).cuda_stream), "ts_counter_add")
    torch.cuda.synchronize(dev)
