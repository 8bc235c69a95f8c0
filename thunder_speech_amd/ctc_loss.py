"""CTC loss on the GPU -- reference signature of src/thunder/ctc_loss.py:15-47.

The reference permutes the logits, takes log_softmax and calls F.ctc_loss(reduction="mean",
zero_infinity=True).  Here one HIP kernel (csrc/ctc.hip, one wavefront per utterance) computes the loss AND
dL/dlogits in the forward call; autograd just scales the stored gradient."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


def _prepare_targets(targets: Tensor, target_lengths: Tensor, b: int, v: int, dev):
    """-> (int32 [B, S_max] padded targets, int32 [B] lengths, bool [B] rows holding a wild id | None) on `dev`.

    F.ctc_loss accepts padded 2-D targets or the 1-D concatenation of all targets (sum(target_lengths) entries); both are
    accepted here, the 1-D form is scattered into the padded layout.  Ids the kernels would index with (positions < length)
    must lie in [0, V).  Host tensors are checked here (ValueError); for device tensors the check stays on the device, without
    a host synchronisation: an utterance with a wild id is made infeasible (its input length becomes 0, so its loss is the
    infinity that zero_infinity turns into 0 and its gradient is 0).  Padding positions are rewritten to 0 either way, so no
    kernel ever indexes with a wild id."""
    tl = target_lengths.to(dtype=torch.int64)
    if targets.dim() == 1:
        tl_host = tl.cpu()
        if int(tl_host.sum()) != targets.numel():
            raise ValueError("calculate_ctc: 1-D targets must hold exactly sum(target_lengths) labels (F.ctc_loss's concatenated format)")
        s_max = max(int(tl_host.max()) if b else 0, 1)
        rows = torch.arange(b).repeat_interleave(tl_host)
        cols = torch.cat([torch.arange(int(n)) for n in tl_host.tolist()]) if targets.numel() else torch.zeros(0, dtype=torch.int64)
        padded = torch.zeros(b, s_max, dtype=torch.int64, device=targets.device)
        padded[rows.to(targets.device), cols.to(targets.device)] = targets.to(torch.int64)
        targets = padded
    elif targets.dim() != 2 or targets.shape[0] != b:
        raise ValueError("calculate_ctc: targets must be [batch, S] or the 1-D concatenation of the target sequences")
    if targets.shape[1] == 0:
        targets = torch.zeros(b, 1, dtype=torch.int64, device=targets.device)
    tg = targets.to(device=dev, dtype=torch.int64)
    tl = tl.to(dev)
    valid = torch.arange(tg.shape[1], device=dev)[None, :] < tl[:, None]
    bad = valid & ((tg < 0) | (tg >= v))
    bad_rows = None
    if not targets.is_cuda:
        if bool(bad.any()):
            raise ValueError(f"calculate_ctc: target ids must lie in [0, {v}) (the number of classes of the logits)")
    else:
        bad_rows = bad.any(dim=1)
    tg = torch.where(valid & ~bad, tg, torch.zeros_like(tg))
    return tg.to(torch.int32).contiguous(), tl.to(torch.int32).contiguous(), bad_rows


_KIND = {torch.int32: 0, torch.int64: 1, torch.float32: 2, torch.float64: 3}

# Device targets cannot be validated without a host synchronisation, so an utterance with a label id outside [0, V) is dropped
# (made infeasible: loss 0, gradient 0) instead of raising as host targets do.  Every such utterance is COUNTED in a device cell per
# GPU; `bad_target_rows()` reads it (one sync) -- call it where the loop synchronises anyway (validation / epoch end): a non-zero value
# means a vocabulary / tokenizer mismatch is silently shrinking the training set.
_BAD_ROWS = {}


def _bad_counter(dev) -> Tensor:
    key = str(torch.device(dev))
    if key not in _BAD_ROWS:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("calculate_ctc: run one eager call before capturing a hipGraph (it creates the persistent bad-target counter)")
        _BAD_ROWS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _BAD_ROWS[key]


def bad_target_rows(device=None, reset: bool = False) -> int:
    """Utterances dropped so far because of out-of-range label ids in DEVICE targets (synchronises)."""
    total = 0
    for key, cell in _BAD_ROWS.items():
        if device is None or str(torch.device(device)) == key:
            total += int(cell.item())
            if reset:
                cell.zero_()
    return total


def _as_kind(t: Tensor, dev, ints_only: bool = False):
    """-> (contiguous tensor on `dev` of a dtype ts_ctc_prepare reads, its kind code)."""
    t = t.to(dev)
    if t.dtype not in _KIND or (ints_only and t.dtype not in (torch.int32, torch.int64)):
        t = t.to(torch.int64)
    return t.contiguous(), _KIND[t.dtype]


def _prepare_on_device(targets: Tensor, target_lengths: Tensor, input_lengths: Tensor, b: int, v: int, dev):
    """_prepare_targets + the input-length handling for 2-D device targets, as ONE launch (ts_ctc_prepare)."""
    if targets.shape[0] != b:
        raise ValueError("calculate_ctc: targets must be [batch, S] or the 1-D concatenation of the target sequences")
    s_in = targets.shape[1]
    s_max = max(s_in, 1)
    tg_in, tk = _as_kind(targets, dev, ints_only=True)
    if s_in == 0:
        tg_in = torch.empty(b, 1, dtype=tg_in.dtype, device=dev)
    tl_in, lk = _as_kind(target_lengths, dev)
    il_in, ik = _as_kind(input_lengths, dev)
    out = torch.empty(b * s_max + 2 * b, dtype=torch.int32, device=dev)
    tg, tl, il = out[: b * s_max].view(b, s_max), out[b * s_max: b * s_max + b], out[b * s_max + b:]
    st = _lib.lib().ts_ctc_prepare(tg_in.data_ptr(), tk, tg_in.stride(0), s_in, tl_in.data_ptr(), lk, il_in.data_ptr(), ik, b, s_max, v,
                                   tg.data_ptr(), tl.data_ptr(), il.data_ptr(), _bad_counter(dev).data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "ts_ctc_prepare")
    return tg, tl, il


class _CtcFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, targets: Tensor, input_lengths: Tensor, target_lengths: Tensor, blank: int):
        if not logits.is_cuda:
            raise RuntimeError("calculate_ctc: GPU tensors required (no CPU fallback)")
        lg = logits.detach()
        b, v, t = lg.shape
        if lg.dtype != torch.float32 or lg.stride(2) != 1 or lg.stride(0) != v * lg.stride(1):
            lg = lg.to(torch.float32).contiguous()
        pitch = lg.stride(1)
        dev = lg.device
        if targets.is_cuda and targets.dim() == 2:
            tg, tl, il = _prepare_on_device(targets, target_lengths, input_lengths, b, v, dev)
        else:
            tg, tl, bad_rows = _prepare_targets(targets, target_lengths, b, v, dev)
            il = input_lengths.to(device=dev, dtype=torch.int64).to(torch.int32).contiguous()     # .long() (A5)
            if bad_rows is not None:
                _bad_counter(dev).add_(bad_rows.sum().to(torch.int32))
                il = torch.where(bad_rows, torch.zeros_like(il), il)
                tl = torch.where(bad_rows & (tl == 0), torch.ones_like(tl), tl)    # keep the utterance infeasible even with an empty target
        s_max = tg.shape[1]
        L = _lib.lib()
        ws = torch.empty(L.ts_ctc_workspace_bytes(b, v, t, s_max), dtype=torch.uint8, device=dev)
        nll = torch.empty(b, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        need_grad = logits.requires_grad
        grad = torch.empty(b, v, pitch, dtype=torch.float32, device=dev) if need_grad else None
        st = L.ts_ctc_loss(lg.data_ptr(), b, v, t, pitch, tg.data_ptr(), s_max, il.data_ptr(), tl.data_ptr(), int(blank),
                           nll.data_ptr(), loss.data_ptr(), grad.data_ptr() if need_grad else None, ws.data_ptr(),
                           torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "ts_ctc_loss")
        ctx.t = t
        ctx.save_for_backward(grad if need_grad else torch.empty(0, device=dev))
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        return g[:, :, : ctx.t] * grad_out, None, None, None, None


def calculate_ctc(probabilities: Tensor, y: Tensor, prob_lengths: Tensor, y_lengths: Tensor, blank_idx: int) -> Tensor:
    """probabilities: [batch, #vocab, time] logits BEFORE softmax; returns the scalar loss (mean over the batch of
    nll / target_length, inf -> 0)."""
    return _CtcFunction.apply(probabilities, y, prob_lengths, y_lengths, blank_idx)
