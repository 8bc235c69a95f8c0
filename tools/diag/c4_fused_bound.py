"""What a fused dw -> pw TRAINING forward launch could gain on the C4 shapes (32 clips x 501 frames), measured instead of priced: the inference
path's fused producer / consumer kernel (csrc/tcs_split.hip, what a train-mode instantiation would be built from; it lacks the store of the
depthwise result `u`, the BatchNorm-in affine of its staging and the per-tile statistics, i.e. it is a LOWER bound of the fused launch) against
the pair the training step launches today (ts_train_dwconv_fwd + the 1x1 product through ts_tcs_subblock_fwd, flags = 0).  Both replayed from a
hipGraph of 20 sub-blocks that ping-pong between two activation buffers, as consecutive layers do."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib, plan, tensors as TS
L = _lib.lib()
B, T = 32, 501


def graph_time(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(side.cuda_stream, 0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(reps):
                fn(side.cuda_stream, i)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (5 * reps) * 1e3)
    return best


for (c, k) in [(256, 33), (256, 39), (512, 51), (512, 63), (512, 75)]:
    torch.manual_seed(k)
    pad = k // 2
    dw = torch.randn(c, 1, k) / k ** 0.5
    pw = torch.randn(c, c, 1) / c ** 0.5
    bn = [torch.rand(c) + 0.5, torch.randn(c) * 0.1, torch.randn(c) * 0.1, torch.rand(c) + 0.5]
    layer = plan.make_tcs_layer("cuda", dw_w=dw, pw_w=pw, bn=bn, kernel=k, stride=1, dilation=1, padding=pad, relu=True)
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    bufs = [TS.arena(("fb", i), B, c, T, "cuda") for i in range(2)]
    for x in bufs:
        x.zero_()
    bufs[0][:, :, :T] = torch.randn(B, c, T, device="cuda").to(torch.bfloat16)
    fused = graph_time(lambda st, i: layer.run(bufs[i & 1], T, lens, out=bufs[1 - (i & 1)], in_tail_zero=True, zero_tail=True))
    # the training pair: depthwise forward (bf16 rows, matrix-core kernel) -> u, then the 1x1 product u -> v
    p = bufs[0].stride(1)
    mid = TS.arena(("fb", 2), B, c, T, "cuda")
    w2 = dw.view(c, k).contiguous().cuda()
    frags, bias = plan.pack_pw_frags(pw.view(c, c).cuda()), torch.zeros(c, device="cuda")
    d = _lib.TcsDesc()
    d.batch, d.c_in, d.c_out, d.t_in, d.t_out, d.pitch_in, d.pitch_out = B, c, c, T, T, p, p
    d.kernel, d.stride, d.dilation, d.padding, d.depthwise, d.relu, d.out_fp32, d.flags = 1, 1, 1, 0, 0, 0, 0, 0
    d.pw_w, d.bias = frags.data_ptr(), bias.data_ptr()

    def pair(st, i):
        x, y = bufs[i & 1], bufs[1 - (i & 1)]
        assert L.ts_train_dwconv_fwd(x.data_ptr(), lens.data_ptr(), None, w2.data_ptr(), mid.data_ptr(), B, c, T, T, k, 1, 1, pad, p, mid.stride(1), 1, st) == 0
        assert L.ts_tcs_subblock_fwd(C.byref(d), mid.data_ptr(), lens.data_ptr(), None, None, y.data_ptr(), st) == 0
    t_pair = graph_time(pair)
    print(f"C={c:4d} K={k:2d}  B={B} T={T}: fused inference kernel {fused:6.1f} us   training pair (dw + 1x1) {t_pair:6.1f} us   bound on the gain {t_pair - fused:5.1f} us", flush=True)
