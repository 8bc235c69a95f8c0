#!/usr/bin/env python3
"""Diagnostic: the same fine-tuning run (QuartzNet15x5, dense tone task, 8 x 2 s, AdamW 1e-3) three ways from one initial state --
CPU autograd through the oracle + torch.optim.AdamW, the HIP path launched eagerly, the HIP path replayed from a hipGraph -- loss per step."""
import copy, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.train_margin_model import build_module, tone_clips

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, SEC = 8, 2.0
ACT = sys.argv[2] if len(sys.argv) > 2 else "fp32"


def data(i, dev):
    return tone_clips(B, SEC, 1000 + i, dev, dense=True)


def run_cpu(m0):
    from oracle import tcs as otcs, frontend as ofe
    torch.set_num_threads(16)
    arch = otcs.quartznet_arch(repeat_blocks=3)
    sd = {k: v.detach().cpu().clone() for k, v in m0.encoder.state_dict().items()}
    dsd = {k: v.detach().cpu().clone() for k, v in m0.decoder.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in list(sd.items()) + list(dsd.items()) if v.is_floating_point() and "running" not in k]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-3)
    out, grads0 = [], None
    for i in range(STEPS):
        wav, l, texts = data(i, "cpu")
        with torch.no_grad():
            feats, fl = ofe.filterbank_features(wav, l.long())
        enc, el = otcs.encoder_forward(arch, sd, feats, fl, training=True)
        logits = otcs.conv1d_decoder_forward(dsd, enc)
        y, yl = m0.text_transform.encode(texts)
        loss = torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), y, el.long(), yl, blank=28, reduction="mean", zero_infinity=True)
        opt.zero_grad(); loss.backward()
        if i == 0:
            grads0 = {k: v.grad.clone() for k, v in list(sd.items()) + [("decoder." + k, v) for k, v in dsd.items()] if getattr(v, "grad", None) is not None}
        opt.step()
        out.append(float(loss.detach()))
    return out, grads0


def run_hip(m0, graphed):
    from thunder_speech_amd import train_ops
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync
    from thunder_speech_amd.train_graph import GraphedTrainStep
    dev = torch.device("cuda", 0)
    m = copy.deepcopy(m0).to(dev).train()
    train_ops.set_activation_dtype(ACT)
    out, grads0 = [], None
    try:
        trainable = [p for p in m.parameters() if p.requires_grad]
        opt, sync = FusedAdamW(trainable, lr=1e-3, weight_decay=1e-3), GradientSync(trainable)
        step = GraphedTrainStep(m, opt, sync, max_target_len=101) if graphed else None
        for i in range(STEPS):
            batch = data(i, dev)
            if graphed:
                loss = step(batch)
            else:
                sync.zero_grad()
                loss = m.training_step(batch, 0)
                loss.backward()
                sync.finish()
                if i == 0:
                    grads0 = {("decoder." + k[len("decoder."):] if k.startswith("decoder.") else k[len("encoder."):]): p.grad.detach().cpu().clone()
                              for k, p in m.named_parameters() if p.grad is not None}
                opt.step()
            out.append(float(loss))
        sync.close()
    finally:
        train_ops.set_activation_dtype("fp32")
    return out, grads0


m0 = build_module("cpu", 0)
t0 = time.time(); hip_e, g_hip = run_hip(m0, False); t1 = time.time()
hip_g, _ = run_hip(m0, True); t2 = time.time()
cpu, g_cpu = run_cpu(m0); t3 = time.time()
print(f"act {ACT}: eager {t1 - t0:.0f} s, graphed {t2 - t1:.0f} s, cpu {t3 - t2:.0f} s")
print("step   cpu-autograd   hip-eager   hip-graphed")
for i in range(STEPS):
    print(f"{i:4d}   {cpu[i]:10.4f}   {hip_e[i]:10.4f}   {hip_g[i]:10.4f}")
worst = []
for k, g in g_cpu.items():
    h = g_hip.get(k)
    if h is None:
        worst.append((float("inf"), k)); continue
    worst.append((float((h - g).norm() / (g.norm() + 1e-12)), k))
worst.sort(reverse=True)
print("step-0 gradients, relative L2 difference hip-eager vs cpu-autograd, worst 8 of", len(worst))
for r, k in worst[:8]:
    print(f"  {r:.3e}  {k}")
