"""C5: wav2vec2-large-960h geometry (random weights), inference 16 x 20 s -- times the HIP encoder path stage by stage.
python tools/bench_c5.py [--batch 16] [--seconds 20] [--layers 24] [--base]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch


def config(base: bool, layers: int):
    h, heads, ffn = (768, 12, 3072) if base else (1024, 16, 4096)
    return SimpleNamespace(conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), hidden_size=h,
                           num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn, num_conv_pos_embeddings=128,
                           num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5, feat_extract_norm="group",
                           do_stable_layer_norm=False, conv_bias=False, hidden_act="gelu", feat_extract_activation="gelu")


def random_state(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=0.02: torch.randn(*s, generator=g) * scale
    sd = {"feature_extractor.conv_layers.0.conv.weight": r(512, 1, 10, scale=0.3),
          "feature_extractor.conv_layers.0.layer_norm.weight": torch.ones(512), "feature_extractor.conv_layers.0.layer_norm.bias": torch.zeros(512)}
    for i, k in enumerate(cfg.conv_kernel[1:], start=1):
        sd[f"feature_extractor.conv_layers.{i}.conv.weight"] = r(512, 512, k, scale=(2.0 / (512 * k)) ** 0.5)
    c = cfg.hidden_size
    sd.update({"feature_projection.layer_norm.weight": torch.ones(512), "feature_projection.layer_norm.bias": torch.zeros(512),
               "feature_projection.projection.weight": r(c, 512), "feature_projection.projection.bias": torch.zeros(c),
               "encoder.pos_conv_embed.conv.weight_g": torch.ones(1, 1, 128), "encoder.pos_conv_embed.conv.weight_v": r(c, c // 16, 128),
               "encoder.pos_conv_embed.conv.bias": torch.zeros(c), "encoder.layer_norm.weight": torch.ones(c), "encoder.layer_norm.bias": torch.zeros(c)})
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}."
        for n in ("q", "k", "v", "out"):
            sd[p + f"attention.{n}_proj.weight"], sd[p + f"attention.{n}_proj.bias"] = r(c, c), torch.zeros(c)
        sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"] = torch.ones(c), torch.zeros(c)
        sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"] = r(cfg.intermediate_size, c), torch.zeros(cfg.intermediate_size)
        sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"] = r(c, cfg.intermediate_size), torch.zeros(c)
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = torch.ones(c), torch.zeros(c)
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=int, default=20)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--base", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--graph", action="store_true", help="replay the forward pass from a hipGraph (no host launch overhead)")
    ap.add_argument("--check", action="store_true", help="time the CPU oracle on clip 0 (16 threads) and compare the HIP output with it")
    a = ap.parse_args()
    from thunder_speech_amd.huggingface.encoder import Wav2Vec2Plan
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    cfg = config(a.base, a.layers)
    sd = random_state(cfg)
    plan = Wav2Vec2Plan(cfg, sd, "cuda", precision=a.precision)
    pre = Wav2Vec2Preprocess()
    x = (0.1 * torch.randn(a.batch, 16000 * a.seconds)).cuda()
    lengths = torch.full((a.batch,), 16000 * a.seconds, dtype=torch.int32, device="cuda")

    def step():
        xn, _ = pre(x, lengths)
        return plan.forward(xn, None)

    with torch.no_grad():
        out = step(); out = step(); torch.cuda.synchronize()
        run = step
        if a.graph:
            g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    gout = step()
            def run():
                g.replay()
                return gout
            run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f = plan.feature_extractor(x); e1.record(); torch.cuda.synchronize()
        fe_ms = e0.elapsed_time(e1)
    t = out.shape[1]
    c, ffn, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    flops_fe = 2 * a.batch * sum(((16000 * a.seconds) // (5 * 2 ** i)) * 512 * 512 * k for i, k in enumerate(cfg.conv_kernel[1:], start=1))
    flops_tr = 2 * a.batch * t * L * (4 * c * c + 2 * c * ffn + 2 * t * c)
    flops_pos = 2 * a.batch * t * c * (c // 16) * 128
    print(f"C5 {'base' if a.base else 'large'} {a.precision} {a.batch}x{a.seconds}s, {L} layers: {dt*1e3:.1f} ms/step -> {a.batch*a.seconds/dt:,.0f} audio-s/s; "
          f"frames {t}; feature extractor {fe_ms:.1f} ms; {(flops_fe+flops_tr+flops_pos)/dt/1e12:.1f} TFLOP/s "
          f"(fe {flops_fe/1e12:.2f} + transformer {flops_tr/1e12:.2f} + pos-conv {flops_pos/1e12:.2f} TFLOP per step); "
          f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    if a.check:
        check_against_oracle(cfg, sd, x[:1].cpu(), out[:1].float().cpu(), a.seconds)


def check_against_oracle(cfg, sd, x0, out0, seconds):
    """CPU oracle (oracle/w2v.py, the restated transformers forward) on ONE clip: parity at full length + the CPU baseline."""
    from oracle import w2v as ow
    torch.set_num_threads(16)
    ocfg = ow.W2VConfig(conv_dim=cfg.conv_dim, conv_kernel=cfg.conv_kernel, conv_stride=cfg.conv_stride, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups)
    xn = (x0 - x0.mean(dim=1, keepdim=True)) / torch.sqrt(x0.var(dim=1, keepdim=True) + 1e-7)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref, _ = ow.forward(ocfg, sd, xn)
        dt = time.perf_counter() - t0
    err = (out0 - ref).abs()
    print(f"CPU oracle (fp32, 16 threads), 1x{seconds}s clip: {dt:.2f} s -> {seconds/dt:.1f} audio-s/s; HIP vs oracle on that clip: "
          f"max |err| {float(err.max()):.4f}, rms {float(err.pow(2).mean().sqrt()):.5f} (outputs are LayerNorm-ed, unit scale)")


if __name__ == "__main__":
    main()
