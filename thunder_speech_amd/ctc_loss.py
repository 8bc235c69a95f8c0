"""CTC loss on the GPU -- reference signature of src/thunder/ctc_loss.py:15-47.

The reference permutes the logits, takes log_softmax and calls F.ctc_loss(reduction="mean",
zero_infinity=True).  Here one HIP kernel (csrc/ctc.hip, one wavefront per utterance) computes the loss AND
dL/dlogits in the forward call; autograd just scales the stored gradient."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


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
        tg = targets.to(device=dev, dtype=torch.int32).contiguous()
        if tg.dim() == 1:
            tg = tg.view(b, -1)
        s_max = tg.shape[1] if tg.numel() else 0
        if s_max == 0:
            tg = torch.zeros(b, 1, dtype=torch.int32, device=dev)
            s_max = 1
        il = input_lengths.to(device=dev, dtype=torch.int64).to(torch.int32).contiguous()     # .long() (A5)
        tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
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
