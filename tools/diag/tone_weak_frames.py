#!/usr/bin/env python3
"""Diagnostic: after training the tone model, which frames of a validation batch have the smallest top-1 / top-2 margin (HIP inference only)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools.train_margin_model import encoder_frames, frame_labels, tone_clips, train

dev = torch.device("cuda", 0)
m, hist = train(dev, verbose=False)
print("loss", hist[-1][1], "validation margin", hist[-1][2])
seed, B, S = 31337, 64, 15
wav, lengths, texts = tone_clips(B, S, seed, dev)
lab = frame_labels(B, encoder_frames(16000 * S), "bursts", torch.Generator().manual_seed(seed))
with torch.no_grad():
    logits, _ = m(wav, lengths)
    top = logits.float().topk(2, dim=1)
margin = (top.values[:, 0] - top.values[:, 1]).cpu()
idx = top.indices.cpu()
flat = margin.flatten().argsort()[:30]
for f in flat.tolist():
    b, t = divmod(f, margin.shape[1])
    ctx = lab[b, max(t - 2, 0): t + 3].tolist()
    print(f"clip {b:2d} frame {t:3d} margin {float(margin[b, t]):6.3f} top2 {idx[b, :, t].tolist()} truth {int(lab[b, t])} context {ctx}")
import collections
weak = (margin < 2.0).nonzero().tolist()
print("frames with margin < 2:", len(weak), "of", margin.numel())
print("by truth label:", sorted(collections.Counter(int(lab[b, t]) for b, t in weak).items()))
print("by position: silent", sum(1 for b, t in weak if lab[b, t] < 0), "burst first", sum(1 for b, t in weak if lab[b, t] >= 0 and lab[b, t - 1] < 0),
      "burst last", sum(1 for b, t in weak if lab[b, t] >= 0 and t + 1 < lab.shape[1] and lab[b, t + 1] < 0))
