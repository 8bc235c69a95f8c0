"""Config C3: Citrinet-1024 inference, batch 32 x 20 s, bf16, one GPU -- a measurement tool, not the bench line.
    python tools/bench_c3.py [--batch 32] [--seconds 20] [--steps 5]
Step = mel front end (80 mels, 25 ms window) + 23 Citrinet blocks (fused sub-block launches + squeeze-excite sequence) +
decoder + greedy decode, replayed from a hipGraph with the inputs resident in HBM."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from thunder_speech_amd.citrinet.compatibility import build_synthetic_citrinet
    from thunder_speech_amd.module import greedy_decode
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    module = build_synthetic_citrinet().to(dev).eval()
    B, S = args.batch, args.seconds
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn(B, 16000 * S, generator=g)).to(dev)
    lengths = torch.full((B,), 16000 * S, dtype=torch.int32, device=dev)

    def step():
        logits, _ = module(wav, lengths)
        return greedy_decode(logits)

    with torch.no_grad():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                step()
        graph.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    gflop = 5051.0 * B / 32 * S / 20          # 5 051 GFLOP per 32 x 20 s batch
    print(f"C3 Citrinet-1024 {B}x{S}s: {dt * 1e3:.2f} ms/step, {B * S / dt:,.0f} audio-s/s, {gflop / dt / 1e3:.0f} TFLOP/s "
          f"({gflop / dt / 1e3 / 2500:.3f} of dense bf16 MFMA peak)")


if __name__ == "__main__":
    main()
