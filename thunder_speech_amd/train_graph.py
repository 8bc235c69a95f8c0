"""One fine-tuning step replayed from a hipGraph (opt-in; no reference counterpart -- the reference leaves the launch loop of
`BaseCTCModule.training_step` (module.py:102-113) + `loss.backward()` to PyTorch/Lightning).

With the encoder unfrozen a QuartzNet15x5 step is ~1 300 short launches (10-40 us each): launched from Python the step is
host-bound.  `GraphedTrainStep` captures everything between the features and the parameter gradients -- encoder, decoder, CTC loss
and the whole backward pass, with the gradients landing in GradientSync's flat bucket buffer -- ONCE per input shape, and replays
it.  Outside the graph, every step: the text encoding (host), the mel front end (two launches; it draws a fresh dither seed and
SpecAugment rectangles per step, which a frozen graph could not), the gradient exchange between ranks (GradientSync.finish) and the
optimizer step (one multi-tensor launch).  Dropout inside the graph stays random across replays through the device-side replay
nonce (rng.bump_replay_nonce, captured at the head of the graph).

    step = GraphedTrainStep(module, optimizer, sync)         # sync = parallel.GradientSync(trainable parameters)
    for batch in loader:
        loss = step(batch)                                   # == training_step + backward + sync.finish + optimizer.step

Targets are padded to `max_target_len` labels (a longer transcript raises); batches of a new (batch, samples) shape capture a new
graph.  The returned loss is the graph's own buffer: read it (`.item()`, `.clone()`) before the next call.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import rng, train_ops
from .ctc_loss import calculate_ctc


class GraphedTrainStep:
    def __init__(self, module, optimizer, sync, max_target_len: int = 512, warmup: int = 2):
        self.module, self.optimizer, self.sync = module, optimizer, sync
        if int(warmup) < 1:
            # the eager passes are what creates the per-stream arena buffers, packs the weight fragments and fills the length caches OUTSIDE
            # the graph's private memory pool; a capture without them would bake one-time work (and pool-owned cache entries) into the graph
            raise ValueError("GraphedTrainStep: warmup must be >= 1 (the capture relies on at least one eager pass)")
        self.max_target_len, self.warmup = int(max_target_len), int(warmup)
        self._graphs: Dict[Tuple, tuple] = {}
        self.replays = 0
        # buffers the captured kernels update through raw pointers: their version counters are bumped after every replay so that
        # `_version`-keyed caches (the BN-folded inference weights) notice
        self._touched = [b for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats
                         for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]

    # ------------------------------------------------------------------------------------------------------------------
    def _body(self, feats, flen, y, ylen):
        m = self.module
        if torch.cuda.is_current_stream_capturing():
            rng.bump_replay_nonce(feats.device)
        self.sync.zero_grad()
        encoded, out_lengths = m.encoder(feats, flen)
        probabilities = m.decoder(encoded)
        loss = calculate_ctc(probabilities, y, out_lengths, ylen, m.text_transform.vocab.blank_idx)
        loss.backward()
        return loss.detach()

    def _capture(self, feats, flen, y, ylen):
        dev = feats.device
        static = (torch.empty_like(feats), torch.empty_like(flen), torch.empty_like(y), torch.empty_like(ylen))
        for s, v in zip(static, (feats, flen, y, ylen)):
            s.copy_(v)
        self.sync.hold(True)                     # no collective inside the graph: the exchange runs after the replay, in finish()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            keep = [b.clone() for b in self._touched]
            for _ in range(self.warmup):         # eager passes: rocBLAS picks its kernels, the allocator reaches its steady state
                self._body(*static)
                self.sync.finish(exchange=False)
            for b, k in zip(self._touched, keep):   # the warm-up passes must not count as training steps: running statistics back
                b.copy_(k)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                loss = self._body(*static)
            self.sync.finish(exchange=False)     # host bookkeeping of the capture pass (the launches above were only recorded)
        torch.cuda.current_stream(dev).wait_stream(side)
        return graph, static, loss

    # ------------------------------------------------------------------------------------------------------------------
    def __call__(self, batch) -> torch.Tensor:
        audio, audio_lengths, texts = batch
        m = self.module
        if not audio.is_cuda:
            raise RuntimeError("GraphedTrainStep: GPU tensors only (no CPU fallback)")
        y, ylen = m.text_transform.encode(texts, device=audio.device)
        if y.shape[1] > self.max_target_len:
            raise ValueError(f"GraphedTrainStep: a transcript of {y.shape[1]} labels exceeds max_target_len={self.max_target_len}")
        y = torch.nn.functional.pad(y, (0, self.max_target_len - y.shape[1]))
        with torch.no_grad():
            feats, flen = m.audio_transform(audio, audio_lengths)          # eager: fresh dither seed / SpecAugment rectangles
        key = (tuple(feats.shape), feats.dtype, str(feats.device), tuple(y.shape))
        if key not in self._graphs:
            self._graphs[key] = self._capture(feats, flen, y, ylen)
        graph, static, loss = self._graphs[key]
        for s, v in zip(static, (feats, flen, y, ylen)):
            s.copy_(v, non_blocking=True)
        graph.replay()
        self.replays += 1
        for b in self._touched:
            torch.autograd.graph.increment_version(b)
        self.sync.finish()                       # ranks > 1: all buckets go out now (bf16 wire, reduce-scatter + all-gather)
        self.optimizer.step()
        train_ops.refresh_weight_copies(self.sync.params)     # the graph reads the weights' bf16 / fragment copies: keep them current
        return loss
