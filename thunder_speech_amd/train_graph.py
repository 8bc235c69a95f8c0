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

`segments=S` (S > 1) replays the step as S graphs instead of one so that the gradient exchange overlaps the backward pass, which a single graph
cannot offer (GradientSync's hooks do not run during a replay): the encoder's blocks are cut into S contiguous stages of about equal gradient
bytes (`split_stages`), the activation that crosses a cut is detached, and the backward pass runs stage by stage from the back -- graph 0 =
forward + loss + backward of the last stage (and the decoder), graph k = backward of stage S-1-k.  `segment_parameters(module, S)` hands
GradientSync the matching buckets (bucket k = the parameters whose gradients graph k completes), and after replaying graph k the step starts
bucket k's reduce-scatter / all-gather on the side stream (`GradientSync.launch`) before it replays graph k + 1 on the compute stream: only the
LAST bucket's exchange (the first stage's parameters) is exposed.  `SegmentedBackward` is that logic without graphs (plain autograd, any device):
the eager warm-up passes run it, and so do the CPU tests.

    sync = GradientSync(trainable, groups=segment_parameters(module, 3, first_share=0.1))
    step = GraphedTrainStep(module, optimizer, sync, segments=3, first_share=0.1)      # the exposed (last) bucket: 10 % of the gradient

Targets are padded to `max_target_len` labels (a longer transcript raises); batches of a new (batch, samples) shape capture a new
graph.  The returned loss is the graph's own buffer: read it (`.item()`, `.clone()`) before the next call.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch

from . import rng, train_ops
from .ctc_loss import calculate_ctc


def _trainable_bytes(mod) -> int:
    return sum(p.numel() * p.element_size() for p in mod.parameters() if p.requires_grad)


def split_stages(blocks: Sequence, n_stages: int, first_share: float = None) -> List[List]:
    """Cut `blocks` (the encoder's children, in forward order) into at most `n_stages` contiguous stages of about equal TRAINABLE parameter
    bytes.  Every stage but the last must own a trainable parameter (its bucket would be empty otherwise), so frozen leading blocks join the
    first stage that trains something; fewer stages come back when the blocks do not allow `n_stages`.
    `first_share` (0 < share < 1): the FIRST stage gets about that share of the bytes and the others split the rest evenly.  The first stage's
    gradients complete last, so its bucket is the one whose exchange nothing overlaps: a small first stage keeps the exposed exchange small."""
    blocks = list(blocks)
    if n_stages < 1 or not blocks:
        raise ValueError("split_stages: need >= 1 stage and >= 1 block")
    sizes = [_trainable_bytes(b) for b in blocks]
    n_stages = max(1, min(n_stages, sum(1 for z in sizes if z > 0)))
    total, stages, cur, acc, done = sum(sizes), [], [], 0, 0
    for i, (blk, sz) in enumerate(zip(blocks, sizes)):
        cur.append(blk)
        acc += sz
        left = n_stages - len(stages) - 1                      # stages still to be opened after the current one
        # close the stage once it has reached its share of what is left, as long as the remaining blocks can still fill the remaining stages
        rest = sum(1 for z in sizes[i + 1:] if z > 0)          # trainable blocks still ahead: each remaining stage needs one
        want = (total - done) / (left + 1)
        if first_share is not None and not stages and n_stages > 1:
            want = float(first_share) * total
        if left > 0 and acc > 0 and rest >= left and (acc >= want or rest == left):
            stages.append(cur)
            done, cur, acc = done + acc, [], 0
    if cur:
        if acc == 0 and stages:
            stages[-1] += cur                                  # trailing frozen blocks: no bucket of their own
        else:
            stages.append(cur)
    return stages


def segment_parameters(module, n_segments: int, first_share: float = None) -> List[List[torch.nn.Parameter]]:
    """The `groups` argument of parallel.GradientSync for `GraphedTrainStep(module, ..., segments=n_segments)`: group k = the trainable
    parameters whose gradients piece k of the segmented backward pass completes -- group 0 the decoder's and the last encoder stage's, the last
    group the first stage's -- each in reverse registration order (the order the gradients land in)."""
    stages = split_stages(list(module.encoder.children()), n_segments, first_share)
    seen, groups = set(), []
    for k, stage in enumerate(reversed(stages)):
        mods = ([module.decoder] if k == 0 else []) + list(reversed(stage))
        grp = []
        for mod in mods:
            for p in reversed(list(mod.parameters())):
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    grp.append(p)
        if grp or k == 0:                                      # split_stages: only a lone stage can be without trainable parameters
            groups.append(grp)
    extra = [p for p in reversed(list(module.parameters())) if p.requires_grad and id(p) not in seen]
    if extra:                                                  # trainable parameters outside encoder / decoder (none in the reference's models)
        groups[0] = extra + groups[0]
    return groups


class SegmentedBackward:
    """Forward through `stages` with the activation detached at every cut, backward in pieces from the back.

        seg = SegmentedBackward(stages, head)       # stages: lists of (x, lengths) -> (x, lengths) modules; head(x, lengths) -> scalar loss
        loss = seg.forward(x, lengths)
        for k in range(seg.n_pieces):
            seg.backward(k)                         # piece 0: loss -> last stage (+ whatever `head` holds); piece k: stage S-1-k
                                                    # ... bucket k of a GradientSync built on segment_parameters is complete here

    The sum of the pieces is exactly `loss.backward()` of the unsegmented model (the chain rule cut at the detached activations)."""

    def __init__(self, stages: Sequence[Sequence], head: Callable):
        self.stages, self.head = [list(s) for s in stages], head
        self.n_pieces = len(self.stages)
        self._cuts, self._loss = [], None

    def forward(self, x, lengths):
        self._cuts = []
        for k, stage in enumerate(self.stages):
            for blk in stage:
                x, lengths = blk(x, lengths)
            if k + 1 < len(self.stages):
                if not x.requires_grad:
                    raise RuntimeError("SegmentedBackward: nothing upstream of a cut requires a gradient (a stage of frozen blocks only)")
                cut = x.detach().requires_grad_(True)
                self._cuts.append((x, cut))
                x = cut
        self._loss = self.head(x, lengths)
        return self._loss

    def backward(self, k: int) -> None:
        if k == 0:
            self._loss.backward()
            return
        out, cut = self._cuts[len(self._cuts) - k]
        g, cut.grad = cut.grad, None
        if g is None:
            raise RuntimeError(f"SegmentedBackward: piece {k} ran before piece {k - 1}")
        out.backward(g)


class GraphedTrainStep:
    def __init__(self, module, optimizer, sync, max_target_len: int = 512, warmup: int = 2, segments: int = 1, first_share: float = None):
        self.module, self.optimizer, self.sync = module, optimizer, sync
        self.segments = 1
        if int(segments) > 1:
            stages = split_stages(list(module.encoder.children()), int(segments), first_share)
            groups = segment_parameters(module, int(segments), first_share)
            want = [sorted(id(p) for p in g) for g in groups]
            have = [sorted(id(sync.params[i]) for i in members) for (_, _, members) in sync.buckets]
            if want != have:
                raise ValueError("GraphedTrainStep(segments=S): build the GradientSync with groups=segment_parameters(module, S) -- "
                                 f"its {len(have)} buckets are not the {len(want)} pieces of the segmented backward pass")
            self.segments, self._stages = len(stages), stages
            # The early per-bucket optimizer step (__call__) updates bucket k's parameters -- and rewrites their bf16 / fragment copies -- on the
            # side stream while graph k + 1 is running.  That is only sound if NO later piece reads a parameter of an earlier bucket: every
            # parameter of bucket k must belong to modules of stage S-1-k (or, for bucket 0, the decoder) and to nothing else.  A parameter shared
            # between stages (tied weights), or one outside encoder / decoder (`extra`, folded into bucket 0), switches the early step off.
            owner = {}
            for si, stage in enumerate(stages):
                for blk in stage:
                    for p in blk.parameters():
                        owner.setdefault(id(p), set()).add(si)
            for p in module.decoder.parameters():
                owner.setdefault(id(p), set()).add(len(stages) - 1)              # the decoder's backward is part of piece 0 = the last stage's
            self._early_ok = True
            for k, grp in enumerate(groups):
                want_stage = len(stages) - 1 - k
                for p in grp:
                    if owner.get(id(p)) != {want_stage}:
                        self._early_ok = False
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
        with train_ops.deferred_wgrad():         # no hook sends a bucket during a replay: the split-K partials of all layers are summed at once
            loss.backward()
        return loss.detach()

    def _pieces(self, feats, flen, y, ylen):
        """The segmented step as a list of callables, one per graph: [forward + loss + backward of the last stage, backward of stage S-2, ...]."""
        from . import tensors as _t
        m = self.module
        head = lambda x, lengths: calculate_ctc(m.decoder(x), y, lengths, ylen, m.text_transform.vocab.blank_idx)
        seg = SegmentedBackward(self._stages, head)
        box = {}

        def first():
            if torch.cuda.is_current_stream_capturing():
                rng.bump_replay_nonce(feats.device)
            self.sync.zero_grad()
            _t.require_gpu(feats, "encoder")
            with _t.lengths_scope():
                box["loss"] = seg.forward(feats, flen)
            with train_ops.deferred_wgrad():     # summed once per piece, before the piece's bucket goes out
                seg.backward(0)
            return box["loss"].detach()

        def later(k):
            with train_ops.deferred_wgrad():
                seg.backward(k)

        return [first] + [(lambda k=k: later(k)) for k in range(1, seg.n_pieces)]

    def _capture_segments(self, feats, flen, y, ylen):
        dev = feats.device
        static = (torch.empty_like(feats), torch.empty_like(flen), torch.empty_like(y), torch.empty_like(ylen))
        for s, v in zip(static, (feats, flen, y, ylen)):
            s.copy_(v)
        self.sync.hold(True)                     # hooks only count: bucket k is started by hand after graph k's replay (__call__)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            keep = [b.clone() for b in self._touched]
            for _ in range(self.warmup):
                for piece in self._pieces(*static):
                    piece()
                self.sync.finish(exchange=False)
            for b, k in zip(self._touched, keep):
                b.copy_(k)
            pool = torch.cuda.graph_pool_handle()            # ONE memory pool: graph k + 1 reads what graph k left (the gradient at the cut)
            graphs, loss = [], None
            for k, piece in enumerate(self._pieces(*static)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=side):
                    out = piece()
                if k == 0:
                    loss = out
                graphs.append(g)
            self.sync.finish(exchange=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        return graphs, static, loss

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
            for _ in range(self.warmup):         # eager passes: weight copies get packed, the allocator reaches its steady state
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
            self._graphs[key] = (self._capture_segments if self.segments > 1 else self._capture)(feats, flen, y, ylen)
        graph, static, loss = self._graphs[key]
        for s, v in zip(static, (feats, flen, y, ylen)):
            s.copy_(v, non_blocking=True)
        early = None
        if self.segments > 1:
            side = self.sync.side_stream() if hasattr(self.sync, "side_stream") else None
            if side is not None and getattr(self.optimizer, "supports_subset", False) and getattr(self, "_early_ok", False):
                early = set()
            for k, g in enumerate(graph):
                g.replay()
                if k + 1 < len(graph):
                    self.sync.launch(k)          # bucket k travels on the side stream while graph k + 1 runs; the last one goes out in finish()
                    if early is not None:
                        # ... and its parameters take their optimizer step right behind it, on the same stream: graph k + 1 (the backward pass of
                        # EARLIER layers) reads none of them, so the update and the re-packing of their weight copies run under it too; only the
                        # last bucket's update is left for after the step (C4, 2 pieces: 90 % of the AdamW + packing launches leave the tail)
                        members = [self.sync.params[i] for i in self.sync.buckets[k][2]]
                        ids = {id(p) for p in members}
                        side.wait_stream(torch.cuda.current_stream(audio.device))     # the bucket is complete (also with the exchange switched off)
                        with torch.cuda.stream(side):
                            self.optimizer.step(only=ids)
                            train_ops.refresh_weight_copies(members)
                        early |= ids
        else:
            graph.replay()
        self.replays += 1
        for b in self._touched:
            torch.autograd.graph.increment_version(b)
        self.sync.finish()                       # ranks > 1: the remaining buckets go out now (bf16 wire, reduce-scatter + all-gather)
        if early:
            rest = {id(p) for p in self.sync.params} - early
            self.optimizer.step(only=rest)
        else:
            self.optimizer.step()
        train_ops.refresh_weight_copies(self.sync.params)     # the graph reads the weights' bf16 / fragment copies: keep them current
        return loss
