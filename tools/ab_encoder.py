"""Same-box A/B of library variants on the C2 encoder: python tools/ab_encoder.py variantA variantB ... ('' = product); each variant runs in
its own process, interleaved twice."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, torch
sys.path.insert(0, %r)
import bench
if os.environ.get("TS_AB_PW_WIDE"):
    from thunder_speech_amd import _lib
    _lib.lib().ts_tcs_pointwise_wide(int(os.environ["TS_AB_PW_WIDE"]))
def time_graph(fn, steps):
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
dev = torch.device("cuda", 0)
m = bench.build_model(dev); m.graph_inference = False
wav = (0.1 * torch.randn(64, 240000, generator=torch.Generator().manual_seed(1234))).to(dev)
ln = torch.full((64,), 240000, dtype=torch.int32, device=dev)
with torch.no_grad():
    f, fl = m.audio_transform(wav, ln)
    m.encoder(f, fl); torch.cuda.synchronize()
    best = min(time_graph(lambda: m.encoder(f, fl), 40) for _ in range(3))
print("RESULT %%.4f" %% best)
''' % ROOT
def main(names):
    res = {n: [] for n in names}
    for rep in range(2):
        for n in names:
            env = dict(os.environ, TS_LIB_VARIANT=n)
            if n.startswith("pw_wide="):   # product library with the pointwise tile switch set (ts_tcs_pointwise_wide)
                env = dict(os.environ, TS_LIB_VARIANT="", TS_AB_PW_WIDE=n.split("=")[1])
            out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, cwd=ROOT)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            res[n].append(float(line[0].split()[1]) if line else float("nan"))
            if not line:
                print(out.stderr[-2000:])
    for n in names:
        print(f"{n or 'product':12s} encoder ms: " + "  ".join(f"{v:.3f}" for v in res[n]), flush=True)
if __name__ == "__main__":
    main(sys.argv[1:] or [""])
