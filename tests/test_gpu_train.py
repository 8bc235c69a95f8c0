"""GPU parity of the fine-tuning path with a frozen encoder (first phase of FinetuneCTCModule): training_step loss,
decoder gradients (ts_decoder_bwd behind ts_ctc_loss) and the fused AdamW step."""
import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import tcs as otcs

pytestmark = pytest.mark.gpu


def _module(seed=0):
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    sd = otcs.synth_encoder_state(arch, seed=seed, calibrate=True)
    dsd = otcs.synth_decoder_state(1024, 29, seed=seed + 1)
    m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=sd, decoder_state=dsd).cuda()
    m.train()                              # Lightning puts the module in train mode ...
    m.encoder.eval()                       # ... and FinetuneEncoderDecoder freezes the encoder (eval mode, no grads)
    for p in m.encoder.parameters():
        p.requires_grad_(False)
    return m, arch, sd, dsd


def test_training_step_loss_and_decoder_gradients_match_cpu_autograd():
    m, arch, sd, dsd = _module()
    g = torch.Generator().manual_seed(5)
    wav = 0.1 * torch.randn(4, 32000, generator=g)
    lengths = torch.tensor([32000.0, 30000.0, 25000.0, 16000.0])
    texts = ["hello world", "the cat", "a", "speech to text"]
    loss = m.training_step((wav.cuda(), lengths.cuda(), texts), 0)
    loss.backward()
    assert m.decoder.weight.grad is not None and all(p.grad is None for p in m.encoder.parameters())
    # CPU oracle: fp32 encoder, torch autograd through the decoder conv and F.ctc_loss
    with torch.no_grad():
        feats, fl = ofe.filterbank_features(wav, lengths)
        enc, el = otcs.encoder_forward(arch, sd, feats, fl)
    w = dsd["weight"].clone().requires_grad_(True)
    b = dsd["bias"].clone().requires_grad_(True)
    logits = torch.nn.functional.conv1d(enc, w, b)
    y, yl = m.text_transform.encode(texts, device="cpu")
    ref = torch.nn.functional.ctc_loss(logits.permute(2, 0, 1).log_softmax(2), y, el.long(), yl,
                                       blank=m.text_transform.vocab.blank_idx, reduction="mean", zero_infinity=True)
    ref.backward()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-2)            # bf16 encoder activations
    gw, gb = m.decoder.weight.grad.cpu(), m.decoder.bias.grad.cpu()
    sw, sb = float(w.grad.abs().max()), float(b.grad.abs().max())
    assert float((gw - w.grad).abs().max()) <= 0.05 * sw
    assert float((gb - b.grad).abs().max()) <= 0.05 * sb
    cos = torch.nn.functional.cosine_similarity(gw.flatten(), w.grad.flatten(), dim=0)
    assert float(cos) > 0.999


def test_decoder_bwd_kernel_is_exact_on_given_inputs():
    """ts_decoder_bwd alone: same bf16 x, same fp32 g -> matches an fp64 einsum to fp32 rounding."""
    from thunder_speech_amd import _lib
    b, v, c, t = 5, 29, 200, 301
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, c, t, generator=g).to(torch.bfloat16)
    gl = torch.randn(b, v, t, generator=g)
    pitch_x = 384
    xb = torch.zeros(b, c, pitch_x, dtype=torch.bfloat16, device="cuda")
    xb[:, :, :t] = x.cuda()
    xb[:, :, t:] = 9.0                                                  # pitch padding must not leak
    gd = gl.cuda().contiguous()
    dw = torch.full((v, c), 7.0, device="cuda")
    db = torch.full((v,), 7.0, device="cuda")
    st = _lib.lib().ts_decoder_bwd(gd.data_ptr(), xb.data_ptr(), b, v, c, t, t, pitch_x, dw.data_ptr(), db.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
    assert st == 0
    ref_w = torch.einsum("bvt,bct->vc", gl.double(), x.double())
    ref_b = gl.double().sum(dim=(0, 2))
    assert float((dw.cpu().double() - ref_w).abs().max()) <= 1e-4 * float(ref_w.abs().max())
    assert float((db.cpu().double() - ref_b).abs().max()) <= 1e-4 * float(ref_b.abs().max())


def test_fused_adamw_matches_torch_adamw():
    from thunder_speech_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(29, 1024, 1, generator=g)
    pa = torch.nn.Parameter(p0.clone().cuda())
    pb = torch.nn.Parameter(p0.clone())
    oa = FusedAdamW([pa], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    ob = torch.optim.AdamW([pb], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    for step in range(5):
        grad = torch.randn(p0.shape, generator=g)
        pa.grad = grad.cuda()
        pb.grad = grad.clone()
        oa.step()
        ob.step()
    assert float((pa.detach().cpu() - pb.detach()).abs().max()) <= 2e-6
    cpu_p = torch.nn.Parameter(torch.zeros(3))
    cpu_p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):                                  # no CPU fallback
        FusedAdamW([cpu_p]).step()


def test_finetune_loop_decreases_loss():
    """A few optimizer steps on one batch with the decoder only: the loss must go down (end-to-end wiring check)."""
    from thunder_speech_amd.optim import FusedAdamW
    m, *_ = _module(seed=2)
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(4, 24000, generator=g)).cuda()
    lengths = torch.full((4,), 24000.0).cuda()
    texts = ["abc", "hello", "data", "test"]
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.0)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = m.training_step((wav, lengths, texts), 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses


def test_graphed_frozen_encoder_gives_the_same_step():
    """graph_frozen_encoder(): front end + frozen encoder replayed from a hipGraph -- the loss is bit-identical to the eager path
    (same kernels, same order) and the decoder gradient equal up to its atomics' summation order, also on the second replay with
    different audio."""
    from oracle import tcs as otcs
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    arch = otcs.quartznet_arch(repeat_blocks=1)
    m = build_synthetic_quartznet(repeat_blocks=1, encoder_state=otcs.synth_encoder_state(arch, seed=0, calibrate=True),
                                  decoder_state=otcs.synth_decoder_state(1024, 29, seed=1)).cuda().train()
    m.encoder.eval()
    m.audio_transform.eval()                           # no dither: the two paths must see the same samples
    for p in m.encoder.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    texts = ["abc", "hello", "data"]
    lengths = torch.tensor([24000.0, 20000.0, 16000.0]).cuda()

    def step(wav):
        m.zero_grad()
        loss = m.training_step((wav, lengths, texts), 0)
        loss.backward()
        return float(loss), m.decoder.weight.grad.clone()

    wavs = [(0.1 * torch.randn(3, 24000, generator=g)).cuda() for _ in range(2)]
    eager = [step(w) for w in wavs]
    m.graph_frozen_encoder()
    graphed = [step(w) for w in wavs]
    for (l0, g0), (l1, g1) in zip(eager, graphed):
        assert l0 == l1                                    # same kernels, same order: the logits and the loss are bit-identical
        torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-6)      # the weight gradient is summed with atomics: order noise only
    assert eager[0][0] != eager[1][0]
