"""Multi-GPU helpers.

Fine-tuning: data parallel, one process per GPU, replicas hold the full weights; the ONE exchange step per
training step is the gradient average.  `GradientSync` does it the way the hardware wants it: flat gradient buffer with
parameter views, buckets launched from autograd hooks while the backward pass is still running, bf16 on the wire,
reduce-scatter + all-gather (RCCL on the GPUs; fp32 + all-reduce over gloo in the CPU tests).  `allreduce_gradients` is the
simple post-backward form (kept for callers that manage their own .grad tensors).  BatchNorm statistics stay per rank, as
under the reference's Lightning DDP (quirk A4).

Inference shards by clip: every clip is an independent unit (eval-mode BatchNorm uses
running statistics; the reference tests batch independence in tests/utils.py:70-97), so each rank owns a
contiguous slice of the clips and there is NO collective on the data path.  The only collectives are the
rendezvous barrier and the max-over-ranks reduction of the measured time (bench.py)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `n_items` for `rank` (first n % world ranks get one extra)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds: float, device=None, force: bool = False) -> float:
    """MAX all-reduce of a host-side duration (RCCL when `device` is a GPU, gloo on CPU).  A one-rank group skips the collective unless
    `force` (bench.py under torch.distributed.run with ONE rank: the multi-rank code path, collective included, on the hardware at hand)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _descendants(pid: int):
    """pids of every live descendant of `pid` (children first), read from /proc: torch's elastic agent starts its workers in their own
    sessions, so a killpg on the launcher's group does not reach them."""
    import os
    kids = {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open(f"/proc/{name}/stat") as f:
                fields = f.read().rsplit(")", 1)[1].split()
            kids.setdefault(int(fields[1]), []).append(int(name))
        except (OSError, IndexError, ValueError):
            continue
    out, todo = [], [pid]
    while todo:
        for k in kids.get(todo.pop(), []):
            out.append(k)
            todo.append(k)
    return out


def launch_ranks(script: str, n_ranks: int, argv, timeout_s: float = None, env=None, stdout=None, stderr=None) -> int:
    """Start `n_ranks` fresh processes of `script` (one per GPU of this node) under `torch.distributed.run` and return the launcher's
    exit code -- what `pl.Trainer(accelerator="gpu", devices=-1)` does for the reference's users (docs/quick reference guide.md:74-79,
    tests/quartznet/test_module_qn.py:46-53: one command starts all ranks).  The children inherit stdout / stderr (or write to the files
    given as `stdout` / `stderr`), so rank 0's result line is the caller's output; a failing rank makes the launcher tear the others down
    and return non-zero.  The ranks are CHILD processes (never an exec of this one), and the parent only waits; the rendezvous is a c10d
    store on a port the launcher picks itself (--rdzv-endpoint 127.0.0.1:0), so concurrent launches cannot collide on a pre-probed port."""
    import os
    import signal
    import subprocess
    import sys
    if n_ranks < 1:
        raise ValueError("launch_ranks: n_ranks must be >= 1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr=127.0.0.1", script] + [str(a) for a in argv]
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this driver (RCCL across processes)
    child_env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=child_env, start_new_session=True, stdout=stdout, stderr=stderr)
    try:
        return proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        # SIGTERM to the agent (it forwards it to its workers and reaps them), a grace period, then SIGKILL to whatever is left of the
        # exact process tree this call started -- the workers live in their own sessions, so they are found through /proc, not the group
        tree = _descendants(proc.pid)
        os.killpg(proc.pid, signal.SIGTERM)
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pass
        for pid in tree + [proc.pid]:
            try:
                os.kill(pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        proc.wait()
        print(f"launch_ranks: {script} gave no result within {timeout_s:.0f} s; its {n_ranks} ranks were stopped", file=sys.stderr, flush=True)
        return 124


class GradientSync:
    """The one exchange step of data-parallel fine-tuning, overlapped with the backward pass.

    What Lightning's DDP does for the reference (module.py:102-127 runs under `Trainer(strategy="ddp")`; BatchNorm statistics
    stay per rank, quirk A4), laid out for xGMI's point-to-point links:
      * ONE flat fp32 gradient buffer; every `p.grad` is a view into it, so autograd accumulates straight into the bucket
        memory -- no torch.cat staging, no copy-back;
      * buckets are filled in REVERSE parameter order (the order gradients become ready) and each bucket's collective is
        launched from a post-accumulate hook the moment its last gradient has landed, on a side stream: it overlaps the rest of
        the backward pass; `finish()` (call it before the optimizer step) makes the compute stream wait for the side stream;
      * the wire follows the precision mode: fp32 by default (what the reference's DDP averages), bf16 when the training path runs with
        bf16 activations (half the bytes over the links; the mean is formed by pre-scaling with 1 / world inside the pack kernel,
        ts_grad_wire_pack); reduce-scatter + all-gather instead of a ring all-reduce, so that RCCL can drive all seven links of a
        GPU at once; few large buckets (default 32 MiB of fp32 gradient each).
    CPU tensors (the gloo tests) take the same bucket / hook logic, by default with fp32 on the wire and a plain all-reduce; asked for
    the GPU branch's form (wire_dtype=bf16, collective="reduce_scatter") they spell the same arithmetic out over gloo, synchronously.

        sync = GradientSync(trainable_parameters)
        loss.backward(); sync.finish(); optimizer.step(); optimizer.zero_grad(set_to_none=False)   # keep the views!
    """

    def __init__(self, params, bucket_bytes: int = 32 << 20, wire_dtype=None, collective: str = None, groups=None, loopback: bool = False):
        """`groups` (optional): the buckets spelled out -- a list of parameter lists in the order their gradients become complete (the
        pieces of a segmented backward pass, train_graph.segment_parameters): bucket k holds exactly groups[k], whatever `bucket_bytes` says.
        `loopback`: with ONE rank, still run every bucket through its pack -> collective -> unpack sequence on the side stream (the
        collective over a one-rank group moves no bytes over links): what bench_extra.c4_ddp times as the exchange's launch + pack cost."""
        import torch.distributed as dist
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradientSync: no trainable parameters")
        self.loopback = bool(loopback)
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("GradientSync: fp32 parameters on one device only")
        self.device = dev
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        on_gpu = dev.type == "cuda"
        if wire_dtype is None:
            # the wire follows the precision mode: fp32 (the reference's DDP averages fp32 gradients) unless the training path runs in
            # its mixed-precision mode (bf16 activations, train_ops.set_activation_dtype), where bf16 halves the bytes on the links
            from . import train_ops
            wire_dtype = torch.bfloat16 if (on_gpu and train_ops.activation_dtype() == torch.bfloat16) else torch.float32
        self.wire_dtype = wire_dtype
        self.collective = collective or ("reduce_scatter" if on_gpu else "all_reduce")
        if self.collective not in ("reduce_scatter", "all_reduce"):
            raise ValueError("collective must be 'reduce_scatter' or 'all_reduce'")
        # ---- buckets over the REVERSED parameter list; every segment starts 16-byte aligned and every bucket's length is a
        # multiple of 8 * world elements (the shards of the reduce-scatter stay 16-byte aligned)
        align = 8 * max(self.world, 1)
        self.buckets = []                       # (start, length, [param indices])
        self._bucket_of = {}
        offsets, off, cur, cur_start = {}, 0, [], 0
        order, cuts = list(reversed(range(len(self.params)))), set()
        if groups is not None:
            index_of = {id(p): i for i, p in enumerate(self.params)}
            order = []
            for grp in groups:
                members = [index_of[id(p)] for p in grp if p.requires_grad and id(p) in index_of]
                if not members:
                    raise ValueError("GradientSync: a group without trainable parameters")
                if order:
                    cuts.add(members[0])
                order += members
            if sorted(order) != list(range(len(self.params))):
                raise ValueError("GradientSync: `groups` must cover every trainable parameter exactly once")
        for idx in order:
            n = self.params[idx].numel()
            if cur and (idx in cuts if groups is not None else (off - cur_start + n) * 4 > bucket_bytes):
                off = -(-off // align) * align
                self.buckets.append((cur_start, off - cur_start, cur))
                cur, cur_start = [], off
            offsets[idx] = off
            cur.append(idx)
            off = -(-(off + n) // 4) * 4
        off = -(-off // align) * align
        self.buckets.append((cur_start, off - cur_start, cur))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for b, (_, _, members) in enumerate(self.buckets):
            for idx in members:
                self._bucket_of[idx] = b
        from . import train_ops
        self._views = []
        for idx, p in enumerate(self.params):
            view = self.flat[offsets[idx]: offsets[idx] + p.numel()].view_as(p)
            self._views.append(view)
            p.grad = view
            train_ops._GRAD_VIEWS[id(p)] = (view, self)  # the training kernels write parameter gradients straight into the bucket
        self._wire = [torch.empty(length, dtype=self.wire_dtype, device=dev) if self.wire_dtype != torch.float32 else None
                      for (_, length, _) in self.buckets]
        self._pending = [len(m) for (_, _, m) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._side = torch.cuda.Stream(device=dev) if on_gpu else None
        self.n_collectives = 0
        self.wire_bytes = 0                     # bytes this rank has put on the wire (per direction), summed over the collectives
        self._hold = False
        self._clean = True                      # the flat buffer holds zeros only (between zero_grad() and the first gradient of the step)
        self._claimed = set()                   # parameters whose bucket view a backward kernel has taken this step (train_ops.grad_out)
        self._zeroed = True                     # zero_grad() ran since the last finish(): untouched views hold zeros, not last step's gradient
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(idx)) for idx, p in enumerate(self.params)]

    # ------------------------------------------------------------------------------------------------------------------
    def _make_hook(self, idx):
        def hook(param):
            view = self._views[idx]
            if param.grad is None:
                return
            if param.grad.data_ptr() != view.data_ptr():
                # a gradient that was produced elsewhere (plain autograd ops, a set_to_none zero_grad): move it into the bucket
                view.copy_(param.grad)
                param.grad = view
            b = self._bucket_of[idx]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._hold:
                self._launch(b)
        return hook

    def _launch(self, b: int) -> None:
        import torch.distributed as dist
        self._launched[b] = True
        if self.world == 1 and not self.loopback:
            return
        start, length, _ = self.buckets[b]
        seg = self.flat[start: start + length]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))      # the bucket's gradients are complete on the compute stream
            ctx = torch.cuda.stream(self._side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        self.wire_bytes += length * (2 if self.wire_dtype == torch.bfloat16 else 4)
        with ctx:
            if self.wire_dtype == torch.float32:
                wire = seg
                wire.mul_(1.0 / self.world)
            else:
                wire = self._wire[b]
                if seg.is_cuda:
                    from . import _lib
                    st = _lib.lib().ts_grad_wire_pack(seg.data_ptr(), wire.data_ptr(), length, 1.0 / self.world, self._side.cuda_stream)
                    _lib.check(st, "ts_grad_wire_pack")
                else:
                    wire.copy_(seg * (1.0 / self.world))      # what ts_grad_wire_pack does: scale in fp32, one rounding to the wire type
            if self.collective == "reduce_scatter" and not seg.is_cuda:
                # CPU (gloo has neither reduce-scatter nor bf16 sums): the same arithmetic, spelled out -- every rank's shard is the
                # sum of the ranks' wire values ROUNDED TO THE WIRE TYPE, and the shards are then gathered
                rank = dist.get_rank()
                total = wire.to(torch.float32)
                dist.all_reduce(total, op=dist.ReduceOp.SUM)
                shard = total.view(self.world, -1)[rank].to(wire.dtype)
                parts = [torch.empty_like(shard) for _ in range(self.world)]
                dist.all_gather(parts, shard)
                wire.copy_(torch.cat(parts))
                self.n_collectives += 2
            elif self.collective == "reduce_scatter":
                if dist.is_available() and dist.is_initialized():
                    shard = wire.view(self.world, -1)[dist.get_rank()]
                    dist.reduce_scatter_tensor(shard, wire, op=dist.ReduceOp.SUM)
                    dist.all_gather_into_tensor(wire, shard)
                self.n_collectives += 2                       # loop-back without a process group: pack + unpack only
            else:
                if dist.is_available() and dist.is_initialized():
                    dist.all_reduce(wire, op=dist.ReduceOp.SUM)
                self.n_collectives += 1
            if wire is not seg:
                if seg.is_cuda:
                    from . import _lib
                    st = _lib.lib().ts_grad_wire_unpack(wire.data_ptr(), seg.data_ptr(), length, self._side.cuda_stream)
                    _lib.check(st, "ts_grad_wire_unpack")
                else:
                    seg.copy_(wire)

    def launch(self, b: int) -> None:
        """Start bucket `b`'s exchange now (side stream, behind everything the compute stream has been given so far).  The explicit form
        of what the hooks do: for a backward pass replayed in pieces from hipGraphs (train_graph.GraphedTrainStep(segments=...)), where no
        hook runs per step -- piece k has just been replayed, bucket k is complete, and piece k + 1 runs while it travels."""
        if not 0 <= b < len(self.buckets):
            raise IndexError(f"GradientSync.launch: no bucket {b}")
        if not self._launched[b]:
            self._launch(b)

    def side_stream(self):
        """The stream the exchange runs on (None on CPU): callers that put more work behind a bucket's exchange -- train_graph.GraphedTrainStep's
        early per-bucket optimizer step -- queue it here."""
        return self._side

    def hold(self, on: bool = True) -> None:
        """While held, the hooks only collect the gradients into the buckets and every collective waits for finish(): the mode of
        a backward pass that is captured into / replayed from a hipGraph (train_graph.GraphedTrainStep), where the host-side hooks
        do not run per step."""
        self._hold = bool(on)

    def finish(self, exchange: bool = True) -> None:
        """After backward(): launch whatever has not been launched (parameters that received no gradient this step keep their
        zeros), then make the compute stream wait for the exchange.  Resets the per-step bookkeeping.  `exchange=False` does the
        bookkeeping only (warm-up / capture passes of a graphed step, whose gradients are discarded)."""
        self._clean = False
        from . import train_ops
        if train_ops._PENDING_ADD:
            # Fork.backward parked a pair of gradients for a block tail whose backward never ran: the sum would be missing from the step
            n = len(train_ops._PENDING_ADD)
            train_ops._PENDING_ADD.clear()
            raise RuntimeError(f"GradientSync.finish: {n} parked gradient pair(s) were never consumed (train_ops.DEFER_FORK_ADD)")
        for p, view in zip(self.params, self._views):
            if p.grad is None:
                if not self._zeroed:
                    view.zero_()                          # a plain optimizer.zero_grad(set_to_none=True) left last step's averaged gradient here
                p.grad = view                             # no gradient this step: the zeros of the bucket
        self._claimed.clear()
        self._zeroed = False
        if exchange:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            if self._side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._pending = [len(m) for (_, _, m) in self.buckets]
        self._launched = [False] * len(self.buckets)

    def zero_grad(self) -> None:
        """One memset of the flat buffer (instead of one per parameter) and `.grad = None` for every parameter: the backward
        kernels of the training path then write their result straight into the bucket views and autograd adopts those tensors
        as `.grad` (train_ops.grad_out) -- no accumulate kernel, no copy.  Parameters that receive no gradient keep the zeros."""
        self.flat.zero_()
        from . import train_ops
        train_ops._PENDING_ADD.clear()                 # gradient pairs a previous, interrupted backward pass may have left parked
        self._clean = True
        self._zeroed = True
        self._claimed.clear()
        for p in self.params:
            p.grad = None

    def claim(self, param) -> bool:
        """True the first time a backward kernel asks for `param`'s bucket view in a step, False after that (see train_ops.grad_out)."""
        k = id(param)
        if k in self._claimed:
            return False
        self._claimed.add(k)
        return True

    def is_clean(self) -> bool:
        """True between zero_grad() and finish(): every bucket view not yet written this step still holds zeros, so a kernel that
        accumulates into one (the depthwise weight gradient) needs no memset of its own."""
        return self._clean

    def close(self) -> None:
        from . import train_ops
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            train_ops._GRAD_VIEWS.pop(id(p), None)


def allreduce_gradients(params, bucket_bytes: int = 64 << 20) -> int:
    """Average `p.grad` of every parameter over the ranks, in place; returns the number of collectives issued.
    Buckets are filled in parameter order up to `bucket_bytes` (default 64 MiB: the whole QuartzNet15x5 gradient, 75.7 MB
    fp32, travels in two all-reduces)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes) \
                and grads[j].dtype == grads[i].dtype and grads[j].device == grads[i].device:
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off: off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        i = j
    return n_coll
