"""Diagnostic: QuartzNet15x5 train-step gradients, HIP vs fp32 oracle autograd, every parameter (worst first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import frontend as ofe, tcs as otcs
from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
rb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gamma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
dgain = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
round_feats = len(sys.argv) > 4 and sys.argv[4] == "bf16feats"
arch = otcs.quartznet_arch(repeat_blocks=rb)
sd = otcs.synth_encoder_state(arch, seed=0, calibrate=True, main_gamma=gamma)
dsd = otcs.synth_decoder_state(1024, 29, seed=1, gain=dgain)
m = build_synthetic_quartznet(repeat_blocks=rb, encoder_state=sd, decoder_state=dsd).cuda().train()
m.audio_transform[0].layer[0].dither = 0.0
rng = np.random.Generator(np.random.PCG64(5))
wav = torch.from_numpy((0.1 * rng.standard_normal((2, 32000))).astype(np.float32)); wav[1, 24000:] = 0
lengths = torch.tensor([32000.0, 24000.0]); texts = ["hello world", "data"]
cap = {}
def hook(name):
    def f(mod, inp, out):
        t = out[0] if isinstance(out, tuple) else out
        cap[name] = t
        if t.requires_grad: t.register_hook(lambda g, n=name: cap.__setitem__(n + "_grad", g.detach().cpu()))
    return f
m.decoder.register_forward_hook(hook("logits")); m.encoder.register_forward_hook(hook("enc"))
for i, blk in enumerate(m.encoder): blk.register_forward_hook(hook(f"blk{i}"))
loss = m.training_step((wav.cuda(), lengths.cuda(), texts), 0); loss.backward()
sd_ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
dref = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
feats, fl = ofe.filterbank_features(wav, lengths)
if round_feats:
    with torch.no_grad():
        f_hip, _ = m.audio_transform(wav.cuda(), lengths.cuda())
    print("features: HIP vs oracle max abs diff", float((f_hip.float().cpu() - feats).abs().max()))
    feats = f_hip.float().cpu()
x, xl = feats, fl
acts = []
for i, spec in enumerate(arch):
    x, xl = otcs.block_forward(spec, sd_ref, f"{i}.", x, xl, training=True)
    x.retain_grad(); acts.append(x)
logits = otcs.conv1d_decoder_forward(dref, x)
logits.retain_grad()
y, yl = m.text_transform.encode(texts)
ref = torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), y, xl.long(), yl, blank=28, reduction="mean", zero_infinity=True)
ref.backward()
print("loss", float(loss), float(ref))
rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
print("logits fwd rel err", rel(cap["logits"].detach().cpu(), logits.detach()), " dL/dlogits rel err", rel(cap["logits_grad"], logits.grad))
for i, a in enumerate(acts):
    print(f"block {i}: fwd rel err {rel(cap[f'blk{i}'].detach().float().cpu(), a.detach()):.2e}  dL/dout rel err {rel(cap[f'blk{i}_grad'], a.grad):.2e}")
for k, p in m.decoder.named_parameters(): print("decoder", k, rel(p.grad.cpu(), dref[k].grad))
rows = []
for k, p in m.encoder.named_parameters():
    want, got = sd_ref[k].grad, p.grad.cpu()
    s = max(float(want.abs().max()), 1e-9)
    rows.append((float((got - want).abs().max()) / s, k, s))
rows.sort(reverse=True)
for r in rows[:12]: print("%.4f %s scale %.3g" % r)
print("median rel err", sorted(r[0] for r in rows)[len(rows)//2])
by_block = {}
for e, k, s in rows: by_block.setdefault(int(k.split(".")[0]), []).append(e)
print({b: round(max(v), 4) for b, v in sorted(by_block.items())})
print("act grad scales", [round(float(a.grad.abs().max()), 4) for a in acts])
