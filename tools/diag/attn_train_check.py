"""Fused training attention (csrc/w2v_attn_train.hip) against the unfused path of huggingface/train.py (materialised probabilities, f32 products), same seed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from thunder_speech_amd import _lib
from thunder_speech_amd.huggingface import train as T
L = _lib.lib()
torch.manual_seed(0)
for (b, t, heads, p, ragged) in [(2, 130, 4, 0.0, False), (3, 499, 4, 0.0, True), (2, 499, 4, 0.1, False), (3, 200, 2, 0.25, True)]:
    c = 64 * heads
    qkv = (torch.randn(b, t, 3 * c, device="cuda") * 1.5).requires_grad_(True)
    key_len = torch.tensor([t, t // 2, 7][:b], dtype=torch.int32, device="cuda") if ragged else None
    seed = 1234567
    ref = T.Attention.apply(qkv, key_len, heads, p, seed)
    dout = torch.randn_like(ref)
    ref.backward(dout)
    dref = qkv.grad.clone(); qkv.grad = None
    q16 = qkv.detach().to(torch.bfloat16).contiguous()
    ctx = torch.empty(b, t, c, device="cuda"); lse2 = torch.empty(b, heads, t, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    wsf = torch.empty(L.ts_w2v_attention_train_fwd_workspace(b, t, c, heads), dtype=torch.uint8, device='cuda')
    rc = L.ts_w2v_attention_train_fwd(q16.data_ptr(), b, t, c, heads, key_len.data_ptr() if key_len is not None else None, p, seed, ctx.data_ptr(), lse2.data_ptr(), wsf.data_ptr(), st)
    torch.cuda.synchronize()
    err = float((ctx - ref).abs().max()) / float(ref.abs().max())
    rel = float((ctx - ref).norm() / ref.norm())
    # lse2 against a direct computation
    q, k, v = qkv.detach().view(b, t, 3, heads, 64).unbind(2)
    s = torch.einsum("bqhd,bkhd->bhqk", q.bfloat16().float(), k.bfloat16().float()) / 8.0
    if key_len is not None:
        lim = torch.where(key_len > 0, key_len, torch.full_like(key_len, t))
        s = s.masked_fill(torch.arange(t, device="cuda")[None, None, None, :] >= lim[:, None, None, None], float("-inf"))
    lse_ref = torch.logsumexp(s, -1) / 0.6931471805599453
    lerr = float((lse2 - lse_ref).abs().max())
    print(f"B={b} T={t} H={heads} p={p} ragged={ragged}: rc {rc}  ctx max err {err:.3e} rel L2 {rel:.3e}  lse2 max err {lerr:.3e}", flush=True)
    if hasattr(L, "ts_w2v_attention_train_bwd"):
        import ctypes as C
        L.ts_w2v_attention_train_bwd_workspace.restype = C.c_int64
        ws = torch.empty(L.ts_w2v_attention_train_bwd_workspace(b, t, c, heads), dtype=torch.uint8, device="cuda")
        dqkv = torch.full_like(qkv.detach(), float("nan"))
        rc = L.ts_w2v_attention_train_bwd(C.c_void_p(q16.data_ptr()), b, t, c, heads, C.c_void_p(key_len.data_ptr() if key_len is not None else None), C.c_float(p), C.c_uint64(seed),
                                          C.c_void_p(dout.data_ptr()), C.c_void_p(ctx.data_ptr()), C.c_void_p(lse2.data_ptr()), C.c_void_p(None), C.c_void_p(dqkv.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_void_p(st))
        torch.cuda.synchronize()
        for name, sl in (("dq", slice(0, c)), ("dk", slice(c, 2 * c)), ("dv", slice(2 * c, 3 * c))):
            g, r = dqkv[..., sl], dref[..., sl]
            print(f"    {name}: rc {rc} rel L2 {float((g - r).norm() / r.norm()):.3e}  max {float((g - r).abs().max()) / float(r.abs().max()):.3e}  finite {bool(torch.isfinite(g).all())}", flush=True)
