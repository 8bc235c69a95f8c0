"""Optimizer step on the GPU: FusedAdamW = torch.optim.AdamW (decoupled weight decay, bias correction, amsgrad off) with
the update done by ts_adamw_step (csrc/train.hip) instead of a chain of ATen elementwise ops.  Pass it as
`optimizer_class=FusedAdamW` to BaseCTCModule / FinetuneCTCModule; the reference's default stays torch.optim.AdamW."""
from __future__ import annotations

import torch

from . import _lib


def _bump_versions(tensors) -> None:
    """The kernels write through raw pointers, which torch cannot see: bump each tensor's version counter so that everything
    keyed on `_version` (the packed / BN-folded inference weights in blocks._PackedCache, autograd's saved-tensor checks)
    notices the update."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    supports_subset = True       # step(only=...) updates a subset of the parameters (train_graph: a bucket's update behind its exchange)

    @torch.no_grad()
    def step(self, closure=None, only=None):
        """`only`: a set of id(parameter) -- update just those (each parameter still takes exactly one update per training step: the caller
        partitions the parameters over its calls).  Runs on the CURRENT stream."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            todo = []
            for p in group["params"]:
                if p.grad is None or (only is not None and id(p) not in only):
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: contiguous fp32 GPU parameters only (no CPU fallback)")
                g = p.grad.to(torch.float32).contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                todo.append((p, g, st))
            if not todo:
                continue
            dev = todo[0][0].device
            stream = torch.cuda.current_stream(dev).cuda_stream
            steps = {st["step"] for _, _, st in todo}
            if len(todo) >= 8 and len(steps) == 1 and all(p.device == dev for p, _, _ in todo):
                # one launch for the whole group: a pointer table (rebuilt every step: gradients are fresh tensors) and one grid
                from .train_ops import bf16_shadow
                shadows = [bf16_shadow(p) for p, _, _ in todo]          # bf16 GEMM operand copies the training path keeps (if any)
                rows = [[p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                         sh[0].data_ptr() if sh is not None else 0] for (p, g, st), sh in zip(todo, shadows)]
                # pinned + non_blocking: the upload is queued behind the step's kernels instead of making the host wait for them
                # (the pinned-memory allocator does not recycle the block before the copy has run)
                table = torch.tensor(rows, dtype=torch.int64, pin_memory=True).to(dev, non_blocking=True)
                rc = L.ts_adamw_multi_step(table.data_ptr(), len(rows), max(r[4] for r in rows), float(group["lr"]), float(b1), float(b2),
                                           float(group["eps"]), float(group["weight_decay"]), int(steps.pop()), stream)
                _lib.check(rc, "ts_adamw_multi_step")
                _bump_versions(p for p, _, _ in todo)
                for (p, _, _), sh in zip(todo, shadows):
                    if sh is not None:
                        sh[1] = p._version                                # the kernel refreshed the copy: it matches the new version
                from .train_ops import refresh_pw_frags
                refresh_pw_frags([p for p, _, _ in todo])                 # MFMA fragment copies of the 1x1-conv weights: one launch
                continue
            for p, g, st in todo:
                rc = L.ts_adamw_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                     p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                     float(group["weight_decay"]), int(st["step"]),
                                     torch.cuda.current_stream(p.device).cuda_stream)
                _lib.check(rc, "ts_adamw_step")
                _bump_versions([p])
        return loss
