"""The other BASELINE.json configurations, measured by the same process as the headline line (bench.py puts the result under
`extra`): C3 Citrinet-1024 inference 32 x 20 s, C4 QuartzNet15x5 fine-tuning (phase 1: frozen encoder, phase 2: everything
trainable) at local batch 32 x 10 s, C5 wav2vec2-large geometry inference 16 x 20 s.  Each entry carries ms/step, the
metric's value, its own roofline object and -- where an oracle run is affordable -- the CPU oracle timed on ONE clip of the
same workload, which doubles as a parity check of the HIP output (max / rms error on that clip).

    python tools/bench_extra.py c3 c4 c5            # stand-alone, prints one JSON object
"""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TF = 2500.0
FP32_PEAK_TF = 157.3
HBM_PEAK_GBS = 8000.0


def _timed(run, steps, warmup=2):
    from tools import prof_mark
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    prof_mark.mark()                                # profiled runs: brackets the timed region in the kernel trace
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof_mark.mark()
    return dt


def _graphed(step, device):
    side, graph = torch.cuda.Stream(device), torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        step()          # the library's arena buffers are per stream: create (and zero) this stream's before the capture, not inside the graph
        with torch.cuda.graph(graph, stream=side):
            out = step()
    return graph.replay, out


def c3(device, batch=32, seconds=20, steps=20, check=True):
    """Citrinet-1024 (reference constructor, SURVEY 8c layer list), bf16 inference; MFMA-bound (5 051 GFLOP / 8.41 GB per batch)."""
    from thunder_speech_amd.citrinet.compatibility import CITRINET_1024_KERNELS, CITRINET_1024_STRIDES, build_synthetic_citrinet
    from thunder_speech_amd.module import greedy_decode
    from thunder_speech_amd.utils import variance_preserving_init_
    torch.manual_seed(0)
    module = build_synthetic_citrinet()
    variance_preserving_init_(module.encoder, module.decoder, seed=0)
    module = module.to(device).eval()
    module.graph_inference = False          # this measurement replays its own graph of the launch sequence
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn(batch, 16000 * seconds, generator=g)).to(device)
    lengths = torch.full((batch,), 16000 * seconds, dtype=torch.int32, device=device)

    def step():
        logits, _ = module(wav, lengths)
        return logits, greedy_decode(logits)

    with torch.no_grad():
        step(); step(); torch.cuda.synchronize()
        replay, out = _graphed(step, device)
        dt = _timed(replay, steps)
    gflop = 5051.0 * batch / 32 * seconds / 20
    gbytes = 8.41 * batch / 32 * seconds / 20
    res = {"workload": f"Citrinet-1024 inference, batch {batch}x{seconds} s, bf16, hipGraph replay (BASELINE.json configs[2])",
           "ms_per_step": dt * 1e3, "value": batch * seconds / dt, "unit": "audio-seconds/s", "steps": steps,
           "roofline": {"bound": "mfma", "achieved": gflop / dt / 1e3, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": gflop / dt / 1e3 / MFMA_BF16_PEAK_TF, "hbm_frac": gbytes / dt / HBM_PEAK_GBS}}
    if check:
        from oracle import frontend as ofe, tcs as otcs
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        arch = otcs.citrinet_arch([1024] * 21, CITRINET_1024_KERNELS, CITRINET_1024_STRIDES, feat_in=80)
        sd = {k: v.detach().cpu() for k, v in module.encoder.state_dict().items()}
        dsd = {k: v.detach().cpu() for k, v in module.decoder.state_dict().items()}
        x0, l0 = wav[:1].cpu(), lengths[:1].cpu()
        with torch.no_grad():
            t0 = time.perf_counter()
            feats, fl = ofe.filterbank_features(x0, l0, ofe.FrontendConfig(n_window_size=400, nfilt=80))
            enc, _ = otcs.encoder_forward(arch, sd, feats, fl)
            ref = otcs.conv1d_decoder_forward(dsd, enc)
            cdt = time.perf_counter() - t0
        got = out[0][:1].float().cpu()
        scale = float(ref.abs().max())
        res["cpu_baseline"] = {"value": seconds / cdt, "unit": "audio-seconds/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"Citrinet-1024 fp32 oracle, 1x{seconds} s clip, one pass"}
        res["check"] = {"vs": "fp32 oracle logits, clip 0, all frames", "max_err_over_scale": float((got - ref).abs().max()) / scale,
                        "rms_err_over_scale": float((got - ref).pow(2).mean().sqrt()) / scale, "logit_scale": scale}
    return res


def c5(device, batch=16, seconds=20, steps=10, check=True):
    """wav2vec2-large-960h geometry (random weights: the checkpoint needs the network), bf16-operand inference; MFMA-bound."""
    from tools.bench_c5 import config, random_state
    from thunder_speech_amd.huggingface.encoder import Wav2Vec2Plan
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    cfg = config(False, 24)
    sd = random_state(cfg)
    plan = Wav2Vec2Plan(cfg, sd, device, precision="bf16")
    pre = Wav2Vec2Preprocess()
    x = (0.1 * torch.randn(batch, 16000 * seconds, generator=torch.Generator().manual_seed(1234))).to(device)
    lengths = torch.full((batch,), 16000 * seconds, dtype=torch.int32, device=device)

    def step():
        xn, _ = pre(x, lengths)
        return plan.forward(xn, None)

    with torch.no_grad():
        step(); step(); torch.cuda.synchronize()
        replay, out = _graphed(step, device)
        dt = _timed(replay, steps)
    t = out.shape[1]
    c, ffn, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    flops = 2 * batch * (sum(((16000 * seconds) // (5 * 2 ** i)) * 512 * 512 * k for i, k in enumerate(cfg.conv_kernel[1:], start=1))
                         + t * L * (4 * c * c + 2 * c * ffn + 2 * t * c) + t * c * (c // 16) * 128)
    res = {"workload": f"wav2vec2-large geometry inference, batch {batch}x{seconds} s, bf16 operands / fp32 accumulation, hipGraph replay "
                       "(BASELINE.json configs[4])",
           "ms_per_step": dt * 1e3, "value": batch * seconds / dt, "unit": "audio-seconds/s", "steps": steps, "frames": t,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / MFMA_BF16_PEAK_TF}}
    if check:
        from oracle import w2v as ow
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        ocfg = ow.W2VConfig(conv_dim=cfg.conv_dim, conv_kernel=cfg.conv_kernel, conv_stride=cfg.conv_stride, hidden_size=c,
                            num_hidden_layers=L, num_attention_heads=cfg.num_attention_heads, intermediate_size=ffn,
                            num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                            num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups)
        x0 = x[:1].cpu()
        xn = (x0 - x0.mean(dim=1, keepdim=True)) / torch.sqrt(x0.var(dim=1, keepdim=True) + 1e-7)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref, _ = ow.forward(ocfg, sd, xn)
            cdt = time.perf_counter() - t0
        err = (out[:1].float().cpu() - ref).abs()
        res["cpu_baseline"] = {"value": seconds / cdt, "unit": "audio-seconds/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"wav2vec2-large fp32 oracle, 1x{seconds} s clip, one pass"}
        res["check"] = {"vs": "fp32 oracle last_hidden_state (LayerNorm-ed, unit scale), clip 0", "max_err": float(err.max()),
                        "rms_err": float(err.pow(2).mean().sqrt())}
    return res


def c5_finetune(device, batch=8, seconds=10, steps=5, train_precision="fp32"):
    """wav2vec2 fine-tuning as the reference runs it (BaseCTCModule.training_step on an HF encoder with the conv feature extractor frozen,
    tests/huggingface/test_module_huggingface.py:33-54): wav2vec2-large geometry (random weights), f32, CTC loss, AdamW on the transformer --
    forward, backward and the optimizer step, eager (autograd nodes over the library's f32 matrix-core GEMM, huggingface/train.py)."""
    import transformers
    from thunder_speech_amd.blocks import linear_decoder
    from thunder_speech_amd.huggingface.encoder import HuggingFaceEncoderAdapt
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    from thunder_speech_amd.module import BaseCTCModule
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    torch.manual_seed(0)
    cfg = transformers.Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, hidden_dropout=0.1,
                                      activation_dropout=0.1, attention_dropout=0.1, feat_proj_dropout=0.1, layerdrop=0.0, mask_time_prob=0.05,
                                      feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False, vocab_size=32)
    enc = HuggingFaceEncoderAdapt(transformers.Wav2Vec2Model(cfg), precision="fp32" if train_precision == "fp32" else "bf16", train_precision=train_precision)
    tokens = [chr(97 + i) for i in range(26)] + [" "]
    module = BaseCTCModule(enc, linear_decoder(1024, len(tokens) + 1, 0.0), Wav2Vec2Preprocess(), BatchTextTransformer(tokens=tokens),
                           optimizer_class=FusedAdamW, optimizer_kwargs={"lr": 1e-5}).to(device).train()
    opt = module.configure_optimizers()
    opt = opt["optimizer"] if isinstance(opt, dict) else opt
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn(batch, 16000 * seconds, generator=g)).to(device)
    lengths = torch.full((batch,), 16000.0 * seconds, device=device)
    texts = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(n),), generator=g)) for n in torch.randint(40, 100, (batch,), generator=g)]

    def step():
        opt.zero_grad()
        loss = module.training_step((wav, lengths, texts), 0)
        loss.backward()
        opt.step()
        return loss.detach()

    first = float(step())
    for _ in range(2 if train_precision == "fp32" else 4):       # warm-up: the caching allocator has seen every size of a step before the timed ones
        step()
    if train_precision != "fp32":
        steps = max(steps, 12)                                   # ~33 ms steps: a single allocator event inside 5 of them once read as 68 ms per step
    dt = _timed(step, steps)
    last = float(step())
    n_train = sum(p.numel() for p in module.parameters() if p.requires_grad)
    t = 16000 * seconds // 320
    c, ffn, L = 1024, 4096, 24
    fwd = 2.0 * batch * t * L * (4 * c * c + 2 * c * ffn + 2 * t * c)               # transformer only: the feature extractor is frozen (forward once)
    how = ("f32, eager autograd over the own f32 GEMM + FusedAdamW" if train_precision == "fp32" else
           "mixed precision (bf16 operands / f32 accumulation in the linear layers' three products on the own bf16 GEMM, in the fused attention forward / backward and in "
           "the positional conv's three products; LayerNorm, softmax arithmetic, master weights, gradients f32), eager autograd + FusedAdamW")
    return {"workload": f"wav2vec2-large geometry fine-tune step (CTC, feature extractor frozen, dropouts + time masking on), batch {batch}x{seconds} s, " + how,
            "train_precision": train_precision, "ms_per_step": dt * 1e3, "value": 1.0 / dt, "unit": "step/s",
            "audio_seconds_per_s": batch * seconds / dt, "steps": steps, "trainable_parameters": n_train, "loss_first_last": [first, last],
            "roofline": {"bound": "mfma", "model": "3 x transformer forward FLOPs / " + ("f32 matrix-core peak (157 TFLOP/s)" if train_precision == "fp32" else "dense bf16 peak"),
                         "achieved": 3.0 * fwd / dt / 1e12, "peak": 157.0 if train_precision == "fp32" else MFMA_BF16_PEAK_TF,
                         "unit": "TFLOP/s", "frac": 3.0 * fwd / dt / 1e12 / (157.0 if train_precision == "fp32" else MFMA_BF16_PEAK_TF)}}


def c4(device, local_batch=32, seconds=10, steps1=30, steps2=30):
    """QuartzNet15x5 fine-tuning, one rank's share of config C4 (global 256 x 10 s over 8 GPUs = local 32).  Phase 1 = the
    reference recipe's first epochs (FinetuneEncoderDecoder: encoder frozen), phase 2 = everything trainable."""
    from thunder_speech_amd import _lib
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.utils import variance_preserving_init_
    out = {}
    g = torch.Generator().manual_seed(1234)
    wav = (0.1 * torch.randn(local_batch, 16000 * seconds, generator=g)).to(device)
    lengths = torch.full((local_batch,), 16000.0 * seconds, device=device)
    texts = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(n),), generator=g)) for n in torch.randint(60, 140, (local_batch,), generator=g)]
    fwd_gflop = 4826.9 * local_batch * seconds / 2560.0             # 4 826.9 GFLOP forward per 256 x 10 s
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.train_graph import GraphedTrainStep
    # phase 2 twice: the mixed-precision path (bf16 activations, f32 master weights / gradients / statistics; fwd+bwd replayed from one
    # hipGraph) is the measured configuration; the f32 path (the reference's arithmetic, what the parity tests check) is reported beside it
    # phase 1 as the reference's schedule leaves the model (callbacks.py:56-62 -> BaseFinetuning.freeze: the encoder's convolutions stop
    # receiving gradients, its BatchNorm parameters keep them, every module stays in train mode), and -- labelled as what it is -- the
    # cheaper variant with the encoder in eval mode and fully frozen (train_batchnorm=False plus an explicit encoder.eval()), where front
    # end + encoder replay from a hipGraph
    from thunder_speech_amd.callbacks import FinetuneEncoderDecoder
    only = os.environ.get("TS_C4_ONLY")             # profiled runs: one variant per process
    for tag, phase, steps, act, graph in (("c4_phase1", 1, steps1, "bf16", True), ("c4_phase1_eval_frozen", 0, steps1, "fp32", False),
                                          ("c4_phase2", 2, steps2, "bf16", True),
                                          ("c4_phase2_fp32", 2, max(steps2 // 3, 5), "fp32", True)):
        if only and tag != only:
            continue
        torch.manual_seed(0)
        m = build_synthetic_quartznet(repeat_blocks=3)
        variance_preserving_init_(m.encoder, m.decoder, seed=0)
        m = m.to(device).train()
        if phase == 1:
            FinetuneEncoderDecoder(train_batchnorm=True).freeze_before_training(m)
        if phase == 0:
            m.encoder.eval()
            for p in m.encoder.parameters():
                p.requires_grad_(False)
            m.graph_frozen_encoder()
        train_ops.set_activation_dtype(act)
        try:
            trainable = [p for p in m.parameters() if p.requires_grad]
            opt, sync = FusedAdamW(trainable, lr=1e-3), GradientSync(trainable)
            graphed = GraphedTrainStep(m, opt, sync, max_target_len=160) if graph else None

            def step():
                if graphed is not None:
                    return graphed((wav, lengths, texts))
                sync.zero_grad()
                loss = m.training_step((wav, lengths, texts), 0)
                loss.backward()
                sync.finish()
                opt.step()
                return loss

            first = float(step())
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            c0 = _lib.CALLS
            step()
            calls = _lib.CALLS - c0
            dt = _timed(step, steps, warmup=0)
            last = float(step())
        finally:
            train_ops.set_activation_dtype("fp32")
        model_flop = (3.0 if phase == 2 else (2.0 if phase == 1 else 1.0)) * fwd_gflop      # phase 1: forward + data gradients; eval-frozen: forward only
        how = ("encoder in eval mode and fully frozen (NOT the reference schedule's state: an explicit encoder.eval()), front end + encoder "
               "hipGraph-replayed, decoder trainable" if phase == 0 else
               "reference schedule's first phase: encoder convolutions frozen, BatchNorm parameters + decoder trainable, encoder in TRAIN mode "
               "(batch statistics); bf16 activations, forward + backward from one hipGraph" if phase == 1 else
               "everything trainable, " + ("bf16 activations (mixed precision: f32 master weights, gradients, BatchNorm statistics), encoder + decoder + "
                                           "CTC + backward replayed from one hipGraph" if act == "bf16" else "f32 activations (the reference's arithmetic), encoder + decoder + CTC + "
                                           "backward replayed from one hipGraph"))
        out[tag] = {
            "workload": f"QuartzNet15x5 fine-tune step (CTC), local batch {local_batch}x{seconds} s, {how} (BASELINE.json configs[3], one rank's share)",
            "ms_per_step": dt * 1e3, "value": 1.0 / dt, "unit": "step/s", "audio_seconds_per_s": local_batch * seconds / dt,
            "steps": steps, "c_abi_calls_per_step": calls, "dtype": act, "loss_first_last": [first, last],
            "roofline": {"bound": "mfma", "model": "3 x forward FLOPs (BASELINE.md section 3)" if phase == 2 else
                                                    ("2 x forward FLOPs (no weight gradients in the encoder)" if phase == 1 else "1 x forward FLOPs (frozen encoder)"),
                         "achieved": model_flop / dt / 1e3, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": model_flop / dt / 1e3 / MFMA_BF16_PEAK_TF, "frac_of_fp32_vector_peak": model_flop / dt / 1e3 / FP32_PEAK_TF}}
        sync.close()
        del m, opt, sync, graphed
        torch.cuda.empty_cache()
    return out


XGMI_LINK_GBS_PER_DIRECTION = 76.5          # 7 point-to-point links per GPU, ~153 GB/s each counting both directions


def _one_rank_group():
    """A one-rank RCCL process group for the loop-back exchange of a single-GPU run (reduce-scatter / all-gather over one rank move no bytes
    over links but cost their launches); None when a group already exists or RCCL cannot come up here."""
    import socket
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized():
        return None
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        return True
    except Exception as e:                                       # noqa: BLE001 -- recorded in the result, the pack / unpack legs are still timed
        print(f"bench_extra.c4_ddp: no one-rank RCCL group ({type(e).__name__}: {e}); loop-back times pack + unpack only", file=sys.stderr, flush=True)
        return None


def c4_ddp(device, world=1, rank=0, global_batch=256, seconds=10, steps=8, segments=2, first_share=0.1):
    """BASELINE.json configs[3] as the reference runs it (Trainer(strategy="ddp"), tests/quartznet/test_module_qn.py:33-53): QuartzNet15x5
    fine-tuning, everything trainable, GLOBAL batch 256 x 10 s split over the ranks (strong scaling: local batch 256 / world), one
    gradient exchange per step (parallel.GradientSync: bf16 wire, reduce-scatter + all-gather over RCCL), FusedAdamW, forward + backward
    replayed from `segments` hipGraphs (train_graph.GraphedTrainStep(segments=2, first_share=0.1): bucket k's exchange runs on the side stream under
    piece k + 1 of the backward pass; only the last bucket -- the first encoder stage's parameters, cut to ~10 % of the gradient bytes -- is exposed;
    two pieces since round 5: the first stage's backward, ~0.9 ms, is longer than the 34 MB bucket needs on the links, and every extra graph costs
    ~0.1 ms of replay gaps -- profiles/round5_c4_segments.txt).  Every rank runs it; the time is the MAX over ranks between two barriers.
    `exchange_ms_exposed` is the step time minus the time of the same step with the exchange switched off (what the links cost after overlap).
    With ONE rank the exchange runs in loop-back (pack -> collective over a one-rank RCCL group -> unpack: every launch of the real exchange, no
    bytes over links) and the local-32 step -- what each of 8 ranks would run -- is timed the same way:
    `projected_speedup_8 = t(local 256, no exchange) / (t(local 32) + exchange_ms_exposed(local 32) + link time of the exposed bucket)`."""
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync, max_over_ranks
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.train_graph import GraphedTrainStep, segment_parameters
    from thunder_speech_amd.utils import variance_preserving_init_
    import torch.distributed as dist
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} does not split over {world} ranks")
    own_group = _one_rank_group() if world == 1 else None
    # a process group is up: bench.py's own (launched under torch.distributed.run, any number of ranks incl. ONE) or the one-rank group above.
    # Then every barrier / max-over-ranks / reduce-scatter / all-gather of the multi-rank path runs, whatever the world size
    group_up = dist.is_available() and dist.is_initialized()

    def measure(local, n_steps):
        g = torch.Generator().manual_seed(1234 + rank)
        wav = (0.1 * torch.randn(local, 16000 * seconds, generator=g)).to(device)
        lengths = torch.full((local,), 16000.0 * seconds, device=device)
        texts = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(n),), generator=g)) for n in torch.randint(60, 140, (local,), generator=g)]
        torch.manual_seed(0)
        m = build_synthetic_quartznet(repeat_blocks=3)
        variance_preserving_init_(m.encoder, m.decoder, seed=0)          # same seed on every rank: replicas start identical
        m = m.to(device).train()
        trainable = [p for p in m.parameters() if p.requires_grad]
        opt = FusedAdamW(trainable, lr=1e-3)
        sync = GradientSync(trainable, groups=segment_parameters(m, segments, first_share), loopback=(world == 1))
        graphed = GraphedTrainStep(m, opt, sync, max_target_len=160, segments=segments, first_share=first_share)
        step = lambda: graphed((wav, lengths, texts))

        def timed(n):
            torch.cuda.synchronize()
            if group_up:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            if group_up:
                dist.barrier()
            return max_over_ranks((time.perf_counter() - t0) / n, device, force=group_up)

        first = float(step())
        for _ in range(4):
            step()
        c0, b0 = sync.n_collectives, sync.wire_bytes
        dt = timed(n_steps)
        n_coll, wire = (sync.n_collectives - c0) // n_steps, (sync.wire_bytes - b0) // n_steps
        # exchange off (every rank steps on its local gradient) and on, in alternating blocks: the exposed exchange is the difference of two
        # ~10 ms measurements, and clock drift between two single blocks is as large as the quantity itself -- the fastest block of each kind counts
        real_world, real_loop = sync.world, sync.loopback
        dt_local = float("inf")
        for _ in range(3):
            sync.world, sync.loopback = 1, False
            dt_local = min(dt_local, timed(max(n_steps // 2, 4)))
            sync.world, sync.loopback = real_world, real_loop
            dt = min(dt, timed(max(n_steps // 2, 4)))
        last = float(step())
        out = {"dt": dt, "dt_local": dt_local, "first": first, "last": last, "n_coll": n_coll, "wire": wire, "wire_dtype": str(sync.wire_dtype).replace("torch.", ""),
               "collective": sync.collective, "n_buckets": len(sync.buckets), "n_graphs": graphed.segments,
               "bucket_bytes_on_wire": [length * (2 if sync.wire_dtype == torch.bfloat16 else 4) for (_, length, _) in sync.buckets]}
        sync.close()
        del m, opt, sync, graphed
        torch.cuda.empty_cache()
        return out

    train_ops.set_activation_dtype("bf16")
    try:
        local = global_batch // world
        r = measure(local, steps)
        r32 = measure(global_batch // 8, 3 * steps) if world == 1 else None
    finally:
        train_ops.set_activation_dtype("fp32")
        if own_group:
            dist.destroy_process_group()
    dt, dt_local = r["dt"], r["dt_local"]
    fwd_gflop = 4826.9 * global_batch * seconds / 2560.0
    res = {"workload": f"QuartzNet15x5 fine-tune step (CTC), GLOBAL batch {global_batch}x{seconds} s over {world} GPU(s) = local {local}, data-parallel, "
                       f"bf16 activations (mixed precision), forward + backward from {r['n_graphs']} hipGraphs (bucket k's exchange under backward piece k + 1), "
                       "GradientSync + FusedAdamW (BASELINE.json configs[3])",
           "n_gpus": world, "scaling": "strong", "ms_per_step": dt * 1e3, "value": 1.0 / dt, "unit": "step/s",
           "audio_seconds_per_s": global_batch * seconds / dt, "steps": steps, "loss_first_last": [r["first"], r["last"]],
           "n_collectives_per_step": r["n_coll"], "wire_bytes_per_step_per_rank": r["wire"], "wire_dtype": r["wire_dtype"],
           "collective": r["collective"] if world > 1 else (f"{r['collective']} over a ONE-rank RCCL group ("
                                                           + ("bench.py's own, launched under torch.distributed.run" if (group_up and not own_group) else "created for this measurement")
                                                           + "): every launch of the real exchange, no bytes over links" if group_up else f"{r['collective']} in loop-back (pack + unpack only)"),
           "process_group": ("torch.distributed.run" if (group_up and not own_group) else ("own one-rank group" if own_group else None)),
           "n_buckets": r["n_buckets"], "n_graphs": r["n_graphs"],
           "ms_per_step_without_exchange": dt_local * 1e3, "exchange_ms_exposed": max(dt - dt_local, 0.0) * 1e3,
           "roofline": {"bound": "mfma", "model": "3 x forward FLOPs of the global batch (BASELINE.md section 3) / (n_gpus x dense bf16 peak)",
                        "achieved": 3.0 * fwd_gflop / dt / 1e3, "peak": MFMA_BF16_PEAK_TF * world, "unit": "TFLOP/s",
                        "frac": 3.0 * fwd_gflop / dt / 1e3 / (MFMA_BF16_PEAK_TF * world)}}
    if r32 is not None:
        # what the links add on an 8-GPU node: only the LAST bucket's reduce-scatter + all-gather is not hidden under a backward piece; direct
        # (all-to-all shaped) exchange over 7 point-to-point links: every rank sends 1/8 of the bucket to each peer over its own link, twice
        exposed_bucket = r32["bucket_bytes_on_wire"][-1]
        link_ms = 2.0 * (exposed_bucket / 8.0) / (XGMI_LINK_GBS_PER_DIRECTION * 1e9) * 1e3
        t32 = r32["dt"] * 1e3
        res.update({"local32_ms_per_step": t32, "local32_ms_per_step_without_exchange": r32["dt_local"] * 1e3,
                    "local32_exchange_ms_exposed_loopback": max(r32["dt"] - r32["dt_local"], 0.0) * 1e3,
                    "exposed_bucket_bytes_on_wire": exposed_bucket, "exposed_link_ms_model": link_ms,
                    "link_model": f"last bucket only; reduce-scatter + all-gather, each 1/8 of the bucket per peer link at {XGMI_LINK_GBS_PER_DIRECTION} GB/s per direction",
                    "projected_speedup_8": dt_local * 1e3 / (t32 + link_ms),
                    "projected_speedup_8_formula": "t(local 256, exchange off) / (t(local 32, loop-back exchange overlapped) + modelled link time of the exposed bucket)"})
    return res


def run(device, which=("c3", "c4", "c5", "c5_finetune"), check=True):
    extra = {}
    for name in which:
        t0 = time.perf_counter()
        try:
            if name == "c3":
                extra["c3"] = c3(device, check=check)
            elif name == "c4":
                extra.update(c4(device))
            elif name == "c5":
                extra["c5"] = c5(device, check=check)
            elif name == "c5_finetune":
                only = os.environ.get("TS_C5FT_ONLY")            # profiled runs: one variant per process
                if only != "bf16":
                    extra["c5_finetune"] = c5_finetune(device)
                if only != "fp32":
                    extra["c5_finetune_bf16"] = c5_finetune(device, train_precision="bf16")
            elif name == "c4_ddp":                   # stand-alone A/B of the segmented step: TS_C4_SEGMENTS / TS_C4_FIRST_SHARE (bench.py runs the defaults)
                extra["c4_ddp"] = c4_ddp(device, segments=int(os.environ.get("TS_C4_SEGMENTS", "2")), first_share=float(os.environ.get("TS_C4_FIRST_SHARE", "0.1")))
        except Exception as e:                      # an extra must never take the headline line down with it: recorded, reported on stderr,
            import traceback                        # and bench.py exits non-zero after printing the (complete) line
            print(f"bench_extra: {name} failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            extra[name] = {"error": f"{type(e).__name__}: {e}"}
        extra.setdefault("_wall_s", {})[name] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()
    return extra


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c3", "c4", "c5", "c5_finetune"]
    print(json.dumps(run(torch.device("cuda", 0), names, check="--no-check" not in sys.argv)))
