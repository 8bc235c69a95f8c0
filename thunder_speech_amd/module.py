"""BaseCTCModule -- the plugin surface of the reference's src/thunder/module.py:25-189.

Subclasses pytorch_lightning.LightningModule when Lightning is installed (it is not in this image), otherwise a
plain nn.Module with the same methods, so `pl.Trainer.fit` keeps working where available."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from . import _lib
from .ctc_loss import calculate_ctc
from .text_processing.transform import BatchTextTransformer

try:  # optional dependency of the reference
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning is absent in the build image
    pl = None
    _Base = nn.Module


def greedy_decode(logits: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """argmax over classes + run-collapse on the GPU.  logits [B, V, T'] fp32 -> (ids [B, T'], collapsed [B, T'],
    counts [B]) int32."""
    if not logits.is_cuda:
        raise RuntimeError("greedy_decode: GPU tensors required (no CPU fallback)")
    b, v, t = logits.shape
    lg = logits
    if lg.dtype != torch.float32 or lg.stride(2) != 1 or lg.stride(0) != v * lg.stride(1):
        lg = lg.to(torch.float32).contiguous()
    ids = torch.empty(b, t, dtype=torch.int32, device=lg.device)
    # collapsed ids and their per-row counts share ONE buffer, so that the host takes them in a single device -> host copy
    # (BatchTextTransformer.decode_collapsed); the views behave like separate tensors for everyone else
    packed = torch.empty(b * t + b, dtype=torch.int32, device=lg.device)
    collapsed, counts = packed[: b * t].view(b, t), packed[b * t:]
    st = _lib.lib().ts_greedy_decode(lg.data_ptr(), b, v, t, lg.stride(1), ids.data_ptr(), collapsed.data_ptr(),
                                     counts.data_ptr(), torch.cuda.current_stream(lg.device).cuda_stream)
    _lib.check(st, "ts_greedy_decode")
    return ids, collapsed, counts


class BaseCTCModule(_Base):
    def __init__(self, encoder: nn.Module, decoder: nn.Module, audio_transform: nn.Module,
                 text_transform: BatchTextTransformer, optimizer_class=torch.optim.AdamW, optimizer_kwargs: Dict = None,
                 lr_scheduler_class=None, lr_scheduler_kwargs: Dict = None, encoder_final_dimension: int = None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.audio_transform = audio_transform
        self.text_transform = text_transform
        self.optimizer_class = optimizer_class
        self.optimizer_kwargs = optimizer_kwargs or {}
        self.lr_scheduler_class = lr_scheduler_class
        self.lr_scheduler_kwargs = lr_scheduler_kwargs or {}
        self.lr_scheduler_interval = self.lr_scheduler_kwargs.pop("interval", "step")
        self.encoder_final_dimension = encoder_final_dimension
        try:  # under Lightning the metric objects must be torchmetrics.Metric instances (self.log takes them)
            if pl is None:
                raise ImportError
            from torchmetrics import CharErrorRate, WordErrorRate
        except Exception:  # otherwise: the same two error rates with the edit distances computed on the device
            from .metrics import CharErrorRate, WordErrorRate
        self.validation_cer, self.validation_wer = CharErrorRate(), WordErrorRate()
        self.example_input_array = (torch.randn((10, 16000)), torch.randint(100, 16000, (10,)))

    # ------------------------------------------------------------------------------------------------------------------
    # Inference as users call it (`module(x, lengths)` / `module.predict(x)` in eval mode under no_grad): the ~90 launches of a forward are
    # replayed from a hipGraph per input signature instead of being issued from Python one by one (no reference counterpart; the reference's
    # forward is module.py:74-86).  `graph_inference`: None = automatic (on for the model families whose whole forward is capture-safe: the
    # mel front end + QuartzNet / Citrinet encoder + this package's decoders), True / False = forced on / off.
    graph_inference: Optional[bool] = None
    MAX_INFERENCE_GRAPHS = 8             # graphs kept per module (outputs + at most one static input copy per signature; the launch arena is shared)

    def _inference_graph_ok(self, x: Tensor) -> bool:
        if not x.is_cuda or self.training or torch.is_grad_enabled() or self.graph_inference is False:
            return False
        if torch.cuda.is_current_stream_capturing():          # the caller is recording its own graph: launch into it
            return False
        if getattr(self, "_frozen_graph", None) is not None:  # graph_frozen_encoder() replays its own graph inside the forward
            return False
        auto = getattr(self, "_graph_auto", None)
        if auto is None:
            from .blocks import _Conv1dDecoder, _LinearDecoder
            from .quartznet.blocks import EncoderSequential
            from .quartznet.transform import _FilterbankFeatures as FilterbankFeatures
            auto = self._graph_auto = (isinstance(self.encoder, EncoderSequential) and isinstance(self.audio_transform, FilterbankFeatures)
                                       and isinstance(self.decoder, (_Conv1dDecoder, _LinearDecoder)))
        if not (self.graph_inference or auto):
            return False
        return not any(m.training for m in (self.encoder, self.decoder, self.audio_transform))

    def _weights_stamp(self):
        """Changes whenever a parameter / buffer of the model is written in place (optimizer step, load_state_dict, manual edits): the sum
        of the tensors' version counters.  The tensor list is rebuilt on train() / _apply() (.to, .half ...) / load_state_dict; code that
        REPLACES a Parameter object calls reset_inference_graphs()."""
        ts = getattr(self, "_stamp_tensors", None)
        if ts is None:
            ts = self._stamp_tensors = [t for m in (self.audio_transform, self.encoder, self.decoder)
                                        for t in list(m.parameters()) + list(m.buffers())]
        return sum(t._version for t in ts)

    def reset_inference_graphs(self) -> None:
        self.__dict__.pop("_stamp_tensors", None)
        self.__dict__.pop("_infer_graphs", None)
        self.__dict__.pop("_graph_auto", None)

    def train(self, mode: bool = True):
        self.reset_inference_graphs()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.reset_inference_graphs()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.reset_inference_graphs()
        return super().load_state_dict(*args, **kwargs)

    def _graphed_inference(self, x: Tensor, lengths: Tensor):
        """(logits, out_lengths, ids, collapsed, counts) -- the graph's OWN output buffers, overwritten by the next call of this signature --
        or None: this input has no graph (yet).  Two kinds of graph per input signature (shapes, dtypes, device):
          * zero-copy: reads the waveform where the caller's tensor lives (keyed by its address; captured the second time a signature is seen
            AT that address -- a serving loop whose batches land in the same allocation, or a few rotating buffers): no input copy per call;
          * copying: one per signature, the waveform is copied into the graph's static buffer first (61 MB at 64 x 15 s, ~1.5 % of a step);
            captured once a signature keeps arriving at new addresses.
        At most MAX_INFERENCE_GRAPHS graphs are kept; everything else runs eagerly (a shape that never repeats would pay three forward passes
        for nothing).  All captures of a module record on ONE side stream, so that re-captures (after a weight update) find the launch arena
        of the previous ones instead of growing a new one (tensors.arena is keyed by stream)."""
        from .utils import GraphedForward
        graphs = self.__dict__.get("_infer_graphs")
        stamp = self._weights_stamp()
        if graphs is None or graphs[0] != stamp:
            def run(xx, ll):
                logits, out_lengths = self._forward_eager(xx, ll)
                return (logits, out_lengths) + greedy_decode(logits)
            streams = self.__dict__.setdefault("_infer_streams", {})
            side = streams.get(str(x.device))
            if side is None:
                side = streams[str(x.device)] = torch.cuda.Stream(device=x.device)
            graphs = self.__dict__["_infer_graphs"] = (stamp, GraphedForward(run, stream=side, alias_first=True), GraphedForward(run, stream=side), {})
        _, gz, gc, seen = graphs
        if lengths.device != x.device:
            lengths = lengths.to(x.device)
        if not x.is_contiguous():
            x = x.contiguous()
        kz, kc = gz.signature(x, lengths), gc.signature(x, lengths)
        if gz.has(kz):
            return gz(x, lengths)
        room = gz.count() + gc.count() < self.MAX_INFERENCE_GRAPHS
        if len(seen) > 512:
            seen.clear()
        nz, nc = seen.get(kz, 0) + 1, seen.get(kc, 0) + 1
        seen[kz], seen[kc] = nz, nc
        if nz >= 2 and room:
            return gz(x, lengths)                         # second sighting of this shape at this address: zero-copy graph
        if gc.has(kc):
            return gc(x, lengths)
        if nc >= 4 and nz < 2 and room:
            return gc(x, lengths)                         # the shape keeps coming back at new addresses: one copying graph for it
        return None

    def forward(self, x: Tensor, lengths: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
        """[batch, time] audio -> (logits [batch, vocab, time'] BEFORE softmax, output lengths)."""
        if self._inference_graph_ok(x):
            out = self._graphed_inference(x, lengths)
            if out is not None:
                return out[0].clone(), out[1].clone()      # the caller may keep them across calls: never hand out the graph's buffers
        return self._forward_eager(x, lengths)

    def _forward_eager(self, x: Tensor, lengths: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
        graphed = getattr(self, "_frozen_graph", None)
        if graphed is not None and x.is_cuda and not self.encoder.training and \
                not any(p.requires_grad for p in self.encoder.parameters()):
            encoded, out_lengths = graphed(x, lengths)             # front end + frozen encoder replayed from a hipGraph
            return self.decoder(encoded), out_lengths
        from . import tensors as _t
        with _t.lengths_scope():       # the front end's frame lengths reach the encoder with their int32 copy already made
            features, feature_lengths = self.audio_transform(x, lengths)
            encoded, out_lengths = self.encoder(features, feature_lengths)
        return self.decoder(encoded), out_lengths

    def graph_frozen_encoder(self, enable: bool = True) -> "BaseCTCModule":
        """Opt-in (no reference counterpart): while the encoder is frozen and in eval mode -- the first phase of the reference's
        fine-tuning recipe -- replay `audio_transform -> encoder` from a hipGraph (one per input shape) instead of launching its
        ~80 kernels from Python every step.  The encoder output then lives in the graph's buffer until the next forward."""
        from .utils import GraphedForward
        self._frozen_graph = GraphedForward(lambda x, lengths: self.encoder(*self.audio_transform(x, lengths))) if enable else None
        return self

    def predict(self, x: Tensor) -> List[str]:
        """Greedy transcription; every clip is treated as full length (module.py:98)."""
        audio_lengths = torch.full((x.shape[0],), x.shape[-1], dtype=torch.int32, device=x.device)
        out = self._graphed_inference(x, audio_lengths) if self._inference_graph_ok(x) else None
        if out is not None:
            collapsed, counts = out[3], out[4]             # consumed right here: no copies
        else:
            pred, _ = self._forward_eager(x, audio_lengths)
            _, collapsed, counts = greedy_decode(pred)
        return self.text_transform.decode_collapsed(collapsed, counts)

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        audio, audio_lengths, texts = batch
        y, y_lengths = self.text_transform.encode(texts, device=audio.device)
        probabilities, prob_lengths = self(audio, audio_lengths)
        loss = calculate_ctc(probabilities, y, prob_lengths, y_lengths, self.text_transform.vocab.blank_idx)
        if pl is not None:
            self.log("loss/train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx: int) -> torch.Tensor:
        audio, audio_lengths, texts = batch
        y, y_lengths = self.text_transform.encode(texts, device=audio.device)
        probabilities, prob_lengths = self(audio, audio_lengths)
        loss = calculate_ctc(probabilities, y, prob_lengths, y_lengths, self.text_transform.vocab.blank_idx)
        _, collapsed, counts = greedy_decode(probabilities)
        decoded_preds = self.text_transform.decode_collapsed(collapsed, counts)
        decoded_targets = self.text_transform.decode_prediction(y, remove_repeated=False)
        self.validation_cer(decoded_preds, decoded_targets)
        self.validation_wer(decoded_preds, decoded_targets)
        if pl is not None:
            self.log("loss/val_loss", loss)
            self.log("metrics/cer", self.validation_cer, on_epoch=True)
            self.log("metrics/wer", self.validation_wer, on_epoch=True)
        return loss

    def _resolve_builder_kwargs(self, kwargs: Dict) -> Dict:
        """The reference's one magic key (module.py:165-171): {"total_steps_arg": name} means "pass the trainer's estimated number of
        optimizer steps under `name`" (e.g. OneCycleLR's total_steps).  Everything else goes to the builder untouched."""
        name = kwargs.get("total_steps_arg")
        resolved = {k: v for k, v in kwargs.items() if k != "total_steps_arg"}
        if name:
            resolved[name] = self.trainer.estimated_stepping_batches
        return resolved

    def _update_special_optimizer_arg(self, kwargs: Dict) -> Dict:
        """The reference's hook name (module.py:165-171) and the one `configure_optimizers` calls: a subclass that overrides it is honoured."""
        return self._resolve_builder_kwargs(kwargs)

    def configure_optimizers(self) -> Union[torch.optim.Optimizer, Dict[str, Any]]:
        """Lightning contract (module.py:173-189): the optimizer over the trainable parameters, alone or with its scheduler entry."""
        trainable = [p for p in self.parameters() if p.requires_grad]
        optimizer = self.optimizer_class(trainable, **self._update_special_optimizer_arg(self.optimizer_kwargs))
        if self.lr_scheduler_class is None:
            return optimizer
        scheduler = self.lr_scheduler_class(optimizer, **self._update_special_optimizer_arg(self.lr_scheduler_kwargs))
        return dict(optimizer=optimizer, lr_scheduler=dict(scheduler=scheduler, interval=self.lr_scheduler_interval))
