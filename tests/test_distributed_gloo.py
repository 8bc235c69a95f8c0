"""world_size-2 gloo test of the N>1 path used by bench.py: clip sharding (no data-path collective) and the
max-over-ranks reduction of the measured time."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thunder_speech_amd.parallel import max_over_ranks, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 65, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_clips, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = shard_range(n_clips, rank, world)
        owned = torch.zeros(n_clips, dtype=torch.int64)
        owned[b:e] = 1
        dist.all_reduce(owned)                       # test-only collective: every clip owned exactly once
        slow = max_over_ranks(1.0 + rank)            # the slowest rank defines the step time
        dist.barrier()
        if rank == 0:
            out.put((owned.tolist(), slow))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_time_reduction():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 65, out)) for r in range(2)]
    for p in procs:
        p.start()
    owned, slow = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert owned == [1] * 65
    assert slow == 2.0


def _grad_worker(rank, world, port, out):
    from thunder_speech_amd.parallel import allreduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)                   # identical "weights" on every rank (replicas)
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((29, 64, 1), (29,), (5, 7), (1000,))]
        params.append(torch.nn.Parameter(torch.zeros(3)))       # a parameter without a gradient is skipped
        for i, p in enumerate(params[:-1]):
            p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
        n_small = allreduce_gradients(params, bucket_bytes=4096)     # forces several buckets
        small = [p.grad.clone() for p in params[:-1]]
        for i, p in enumerate(params[:-1]):
            p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
        n_big = allreduce_gradients(params)                          # one bucket
        dist.barrier()
        if rank == 0:
            out.put((n_small, n_big, [float(s.mean()) for s in small], [float(p.grad.mean()) for p in params[:-1]],
                     params[-1].grad is None))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_averages_in_buckets():
    """The one exchange step of data-parallel fine-tuning: mean of the replicas' gradients."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    n_small, n_big, small, big, untouched = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_big == 1 and n_small > 1 and untouched
    want = [1.5 * (i + 1) for i in range(4)]                   # mean of (rank + 1) * (i + 1) over ranks 0, 1
    assert small == pytest.approx(want) and big == pytest.approx(want)


# ----------------------------------------------------------------------------------------------------------------------
# GradientSync: flat buffer + hooks + buckets (the overlapped exchange of SURVEY 8e), and the statement it has to satisfy:
# n ranks x local batch b == one process averaging the gradients of n independent b-batches (per-rank BatchNorm statistics,
# quirk A4; the reference's DDP test is tests/quartznet/test_module_qn.py:33-53).
# ----------------------------------------------------------------------------------------------------------------------
def _tiny_model_state():
    """A 2-block QuartzNet-style stack + decoder evaluated by the ORACLE (CPU, train-mode BatchNorm): the gloo test needs a
    model that runs without a GPU; the product modules refuse CPU tensors by design."""
    from oracle import tcs as otcs
    arch = [otcs.BlockSpec(8, 16, repeat=2, kernel=5), otcs.BlockSpec(16, 16, repeat=1, kernel=3, residual=False)]
    sd = otcs.synth_encoder_state(arch, seed=3)
    dsd = otcs.synth_decoder_state(16, 6, seed=4)
    return arch, sd, dsd


def _tiny_loss(arch, params, buffers, dparams, rank_seed):
    from oracle import tcs as otcs
    g = torch.Generator().manual_seed(100 + rank_seed)
    x = torch.randn(3, 8, 40, generator=g)
    lengths = torch.tensor([40, 33, 21])
    sd = dict(buffers)
    sd.update(params)
    h, hl = x, lengths
    for i, spec in enumerate(arch):
        h, hl = otcs.block_forward(spec, sd, f"{i}.", h, hl, training=True)          # batch statistics of THIS rank's clips only
    logits = otcs.conv1d_decoder_forward(dparams, h)
    targets = torch.randint(0, 5, (3, 4), generator=g)
    return torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), targets, hl.long(), torch.tensor([4, 3, 2]), blank=5,
                                        reduction="mean", zero_infinity=True)


def _split(sd):
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    return params, buffers


def _sync_worker(rank, world, port, out, hold=False):
    from thunder_speech_amd.parallel import GradientSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        arch, sd, dsd = _tiny_model_state()
        params, buffers = _split(sd)
        dparams = {k: torch.nn.Parameter(v.clone()) for k, v in dsd.items()}
        every = list(params.values()) + list(dparams.values())
        sync = GradientSync(every, bucket_bytes=2048)                 # several buckets
        sync.hold(hold)                                               # held (a graphed step): the hooks only collect, finish() exchanges
        opt = torch.optim.AdamW(every, lr=1e-2)
        launched_during_backward = []
        for step in range(2):
            sync.zero_grad()
            loss = _tiny_loss(arch, params, buffers, dparams, rank + 10 * step)
            loss.backward()
            launched_during_backward.append(sum(sync._launched))         # buckets already exchanged by the hooks, before finish()
            sync.finish()
            opt.step()
        dist.barrier()
        out.put((rank, {k: v.detach().numpy().copy() for k, v in list(params.items()) + [("dec." + k, v) for k, v in dparams.items()]},
                 len(sync.buckets), launched_during_backward, sync.n_collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hold", [False, True])
def test_two_rank_training_step_equals_one_process_averaging_two_independent_batches(hold):
    """hold=True is the mode train_graph.GraphedTrainStep puts GradientSync in (no collective while the backward pass is captured /
    replayed; every bucket goes out in finish()): same result, nothing launched from the hooks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, out, hold)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(2):
        rank, state, n_buckets, launched, n_coll = out.get(timeout=300)
        results[rank] = (state, n_buckets, launched, n_coll)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process: two independent local batches per step (each with its OWN BatchNorm statistics), gradients averaged
    torch.set_num_threads(1)
    arch, sd, dsd = _tiny_model_state()
    params, buffers = _split(sd)
    dparams = {k: torch.nn.Parameter(v.clone()) for k, v in dsd.items()}
    every = list(params.values()) + list(dparams.values())
    opt = torch.optim.AdamW(every, lr=1e-2)
    for step in range(2):
        grads = [torch.zeros_like(p) for p in every]
        for rank in range(2):
            for p in every:
                p.grad = None
            _tiny_loss(arch, params, dict(buffers), dparams, rank + 10 * step).backward()
            for g, p in zip(grads, every):
                g += p.grad / 2
        for g, p in zip(grads, every):
            p.grad = g
        opt.step()
    want = {k: v.detach() for k, v in list(params.items()) + [("dec." + k, v) for k, v in dparams.items()]}
    for rank in (0, 1):
        state, n_buckets, launched, n_coll = results[rank]
        assert n_buckets > 2 and n_coll == 2 * n_buckets                     # one all-reduce per bucket per step (gloo path)
        assert all(n == (0 if hold else n_buckets) for n in launched)        # every bucket went out from a hook, during backward (or none, held)
        for k, v in want.items():
            got = torch.from_numpy(state[k])
            assert torch.allclose(got, v, rtol=1e-5, atol=1e-6), (rank, k, float((got - v).abs().max()))
    for k in want:                                                           # replicas stay bit-identical
        assert (results[0][0][k] == results[1][0][k]).all(), k


def _wire_worker(rank, world, port, out):
    """The GPU branch's exchange arithmetic on CPU tensors: bf16 wire with the 1 / world scaling inside the pack, reduce-scatter +
    all-gather (spelled out over gloo), unpack to fp32."""
    from thunder_speech_amd.parallel import GradientSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in (1000, 37, 4096, 5)]
        sync = GradientSync(params, bucket_bytes=8192, wire_dtype=torch.bfloat16, collective="reduce_scatter")
        sync.zero_grad()
        gr = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(p.shape, generator=gr) for p in params]
        loss = sum((p * gg).sum() for p, gg in zip(params, grads))            # dL/dp = this rank's random gradient
        loss.backward()
        sync.finish()
        out.put((rank, [p.grad.detach().clone().numpy() for p in params], sync.n_collectives, len(sync.buckets), sync.wire_bytes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bf16_wire_reduce_scatter_all_gather_arithmetic_at_world_2():
    """ts_grad_wire_pack / unpack + reduce-scatter + all-gather, as GradientSync runs them on the GPUs (parallel.py _launch), checked
    at world 2 over gloo: the averaged gradient is bf16(bf16(g0 / 2) + bf16(g1 / 2)) -- one rounding per rank in the pack, one in the
    bf16 sum -- on EVERY rank bit for bit, and within bf16 resolution of the fp32 mean."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_wire_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, grads, n_coll, n_buckets, wire_bytes = out.get(timeout=300)
        res[rank] = (grads, n_coll, n_buckets, wire_bytes)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shapes = (1000, 37, 4096, 5)
    local = []
    for rank in range(2):
        gr = torch.Generator().manual_seed(100 + rank)
        local.append([torch.randn(n, generator=gr) for n in shapes])
    for i, n in enumerate(shapes):
        a, b = local[0][i], local[1][i]
        want = ((a * 0.5).to(torch.bfloat16).float() + (b * 0.5).to(torch.bfloat16).float()).to(torch.bfloat16).float()
        for rank in range(2):
            got = torch.from_numpy(res[rank][0][i])
            assert torch.equal(got, want), (rank, i, float((got - want).abs().max()))
        mean = (a + b) / 2
        assert float((want - mean).abs().max()) <= 2 ** -7 * float(mean.abs().max())
    n_buckets = res[0][2]
    assert n_buckets >= 2 and res[0][1] == 2 * n_buckets                       # reduce-scatter + all-gather per bucket
    assert res[0][3] == res[1][3] and res[0][3] >= 2 * sum(shapes)             # bf16: two bytes per gradient element on the wire


# ----------------------------------------------------------------------------------------------------------------------
# Segmented backward pass (train_graph.SegmentedBackward / segment_parameters / GradientSync(groups=...).launch): the mode
# GraphedTrainStep(segments=S) replays from S hipGraphs so that bucket k's exchange runs under piece k + 1.  Plain torch
# modules stand in for the encoder blocks here (same (x, lengths) -> (x, lengths) signature); the graphs themselves need a GPU
# (tests/test_gpu_train_graph.py).
# ----------------------------------------------------------------------------------------------------------------------
class _Blk(torch.nn.Module):
    def __init__(self, c_in, c_out, k, trainable=True):
        super().__init__()
        self.conv = torch.nn.Conv1d(c_in, c_out, k, padding=k // 2, bias=False)
        self.bn = torch.nn.BatchNorm1d(c_out)
        if not trainable:
            for p in self.parameters():
                p.requires_grad_(False)

    def forward(self, x, lengths):
        return torch.relu(self.bn(self.conv(x))), lengths


class _Net(torch.nn.Module):
    def __init__(self, seed=0, frozen_first=False):
        super().__init__()
        torch.manual_seed(seed)
        self.encoder = torch.nn.Sequential(_Blk(4, 8, 3, trainable=not frozen_first), _Blk(8, 8, 5), _Blk(8, 16, 3), _Blk(16, 16, 3), _Blk(16, 8, 1))
        self.decoder = torch.nn.Conv1d(8, 5, 1)


def _net_batch(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, 4, 24, generator=g), torch.tensor([24, 20, 11]), torch.randn(3, 5, 24, generator=g)


def _net_head(net, target):
    return lambda x, lengths: ((net.decoder(x) - target) ** 2).mean()


def test_split_stages_balances_trainable_bytes_and_segment_parameters_orders_by_completion():
    from thunder_speech_amd.train_graph import segment_parameters, split_stages
    net = _Net()
    blocks = list(net.encoder.children())
    stages = split_stages(blocks, 3)
    assert [b for st in stages for b in st] == blocks and len(stages) == 3 and all(st for st in stages)
    groups = segment_parameters(net, 3)
    assert len(groups) == 3
    assert groups[0][0] is net.decoder.bias and groups[0][1] is net.decoder.weight           # the decoder's gradients land first
    assert groups[-1][-1] is blocks[0].conv.weight                                               # the first layer's weight lands last
    flat = [id(p) for g in groups for p in g]
    assert sorted(flat) == sorted(id(p) for p in net.parameters()) and len(set(flat)) == len(flat)
    # a frozen leading block joins the first stage that trains something; its parameters are in no group
    net2 = _Net(frozen_first=True)
    st2 = split_stages(list(net2.encoder.children()), 4)
    assert st2[0][0] is net2.encoder[0] and len(st2[0]) >= 2 and all(any(p.requires_grad for b in st for p in b.parameters()) for st in st2)
    assert all(p.requires_grad for g in segment_parameters(net2, 4) for p in g)
    assert len(split_stages(blocks, 1)) == 1 and len(split_stages(blocks, 9)) == 5              # never more stages than trainable blocks
    # first_share: a small FIRST stage (its bucket is the one no backward piece overlaps), the rest split evenly
    small = split_stages(blocks, 3, first_share=0.05)
    assert small[0] == blocks[:1] and [b for st in small for b in st] == blocks and len(small) == 3
    g_small = segment_parameters(net, 3, first_share=0.05)
    assert sum(p.numel() for p in g_small[-1]) < sum(p.numel() for p in groups[-1])


def test_segmented_backward_equals_plain_backward():
    from thunder_speech_amd.train_graph import SegmentedBackward, split_stages
    net, ref = _Net(seed=1), _Net(seed=1)
    x, lengths, target = _net_batch(5)
    h = x
    for blk in ref.encoder:
        h, _ = blk(h, lengths)
    _net_head(ref, target)(h, lengths).backward()
    seg = SegmentedBackward(split_stages(list(net.encoder.children()), 3), _net_head(net, target))
    loss = seg.forward(x, lengths)
    with pytest.raises(RuntimeError):
        seg.backward(2)                                       # pieces run from the back
    done = []
    for k in range(seg.n_pieces):
        seg.backward(k)
        done.append([n for n, p in net.named_parameters() if p.grad is not None])
    assert len(done[0]) < len(done[1]) < len(done[2])         # each piece completes the gradients of one more stage
    assert torch.isfinite(loss)
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), n


def _segment_worker(rank, world, port, out):
    from thunder_speech_amd.parallel import GradientSync
    from thunder_speech_amd.train_graph import SegmentedBackward, segment_parameters, split_stages
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        net = _Net(seed=2)
        trainable = [p for p in net.parameters() if p.requires_grad]
        sync = GradientSync(trainable, groups=segment_parameters(net, 3))
        sync.hold(True)                                       # as under a graph replay: the hooks never start a collective
        opt = torch.optim.SGD(trainable, lr=0.1)
        log = []
        for step in range(2):
            x, lengths, target = _net_batch(50 + rank + 10 * step)
            seg = SegmentedBackward(split_stages(list(net.encoder.children()), 3), _net_head(net, target))
            sync.zero_grad()
            seg.forward(x, lengths)
            for k in range(seg.n_pieces):
                seg.backward(k)
                if k + 1 < seg.n_pieces:
                    sync.launch(k)                            # bucket k goes out BEFORE piece k + 1 runs
                log.append((k, list(sync._launched), sync.n_collectives))
            sync.finish()
            opt.step()
        dist.barrier()
        out.put((rank, {n: p.detach().numpy().copy() for n, p in net.named_parameters()}, len(sync.buckets), log, sync.n_collectives))
    finally:
        dist.destroy_process_group()


def test_two_rank_segmented_step_overlaps_bucket_k_with_piece_k_plus_1_and_averages_exactly():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_segment_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, state, n_buckets, log, n_coll = out.get(timeout=300)
        results[rank] = (state, n_buckets, log, n_coll)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.set_num_threads(1)
    net = _Net(seed=2)
    params = list(net.parameters())
    opt = torch.optim.SGD(params, lr=0.1)
    for step in range(2):
        grads = [torch.zeros_like(p) for p in params]
        for rank in range(2):
            x, lengths, target = _net_batch(50 + rank + 10 * step)
            for p in params:
                p.grad = None
            h = x
            for blk in net.encoder:
                h, _ = blk(h, lengths)
            _net_head(net, target)(h, lengths).backward()
            for g, p in zip(grads, params):
                g += p.grad / 2
        for g, p in zip(grads, params):
            p.grad = g
        opt.step()
    for rank in (0, 1):
        state, n_buckets, log, n_coll = results[rank]
        assert n_buckets == 3 and n_coll == 2 * 3                      # one collective per bucket per step (gloo: all-reduce)
        # after piece k of a step: buckets 0..k-1... launched exactly the ones whose piece is done, except the last (it goes out in finish())
        for i, (k, launched, _) in enumerate(log):
            assert launched == [j <= k and j < 2 for j in range(3)], (i, k, launched)
        for n, p in net.named_parameters():
            got = torch.from_numpy(state[n])
            assert torch.allclose(got, p.detach(), rtol=1e-5, atol=1e-6), (rank, n, float((got - p.detach()).abs().max()))
    for n in results[0][0]:
        assert (results[0][0][n] == results[1][0][n]).all(), n


# ---- launch_ranks: one command starts all ranks (bench.py --gpus N without a launcher) -------------------------------------------------
_STUB = '''
import json, os, sys
import torch, torch.distributed as dist
mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail" and rank == 1:
    print("stub: rank 1 has no GPU", file=sys.stderr); sys.exit(2)
if mode == "hang":
    import time; time.sleep(600)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.ones(1); dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"world_seen_by_all_reduce": int(t.item()), "argv": sys.argv[1:]}), flush=True)
dist.barrier(); dist.destroy_process_group()
'''


def _run_launcher(tmp_path, mode, timeout_s=None):
    """launch_ranks from a fresh interpreter (as bench.py calls it), stdout captured."""
    import subprocess
    import sys
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); from thunder_speech_amd.parallel import launch_ranks; "
            f"sys.exit(launch_ranks({str(stub)!r}, 2, [{mode!r}, '--steps', 3], timeout_s={timeout_s!r}))")
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


def test_launch_ranks_starts_both_ranks_and_relays_rank0_line(tmp_path):
    import json
    r = _run_launcher(tmp_path, "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got == {"world_seen_by_all_reduce": 2, "argv": ["ok", "--steps", "3"]}


def test_launch_ranks_reports_a_failed_rank_with_a_nonzero_code_and_does_not_hang(tmp_path):
    r = _run_launcher(tmp_path, "fail")
    assert r.returncode != 0
    assert "rank 1 has no GPU" in r.stderr


def test_launch_ranks_stops_its_own_process_group_on_timeout(tmp_path):
    r = _run_launcher(tmp_path, "hang", timeout_s=20)
    assert r.returncode == 124
    assert "ranks were stopped" in r.stderr


def test_bench_gpus_2_without_a_launcher_fails_loudly_on_a_gpu_less_box():
    """`python bench.py --gpus 2` starts two child ranks itself; without a GPU both say so and the command exits non-zero."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("GPU box: the real run is the driver's")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("no GPU visible") >= 1          # the launcher may stop the second rank before it has spoken
