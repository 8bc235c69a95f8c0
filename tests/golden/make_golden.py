"""Generate the golden fixtures in this directory from the REAL reference modules.

Runs only in the build container (needs /root/reference, read-only).  The reference's Python is
imported in place -- nothing from it is copied -- following the recipe in SURVEY.md Appendix B:
a package object for `thunder` whose __path__ points at /root/reference/src/thunder (bypasses the
dist-metadata lookup in src/thunder/__init__.py:1-6), and a stand-in for the two
`torchaudio.functional` symbols the front end imports (torchaudio is absent from this image; the
stand-in `melscale_fbanks` is our own slaney restatement, cross-checked against
transformers.audio_utils.mel_filter_bank in tests/test_oracle_frontend.py).

Every fixture stores INPUTS (incl. the exact weights that were loaded into the reference module) and
the reference's OUTPUTS.  While generating, each output is also compared with the oracle restatement
(oracle/*.py) and the script aborts on mismatch, so a committed fixture certifies
"oracle == reference" at generation time; tests/test_oracle_*.py re-check oracle vs fixture without
the reference.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types
import warnings
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/thunder"

warnings.filterwarnings("ignore")


def import_reference():
    import transformers  # noqa: F401  must precede the torchaudio stand-in
    from oracle.frontend import slaney_mel_filterbank

    pkg = types.ModuleType("thunder")
    pkg.__path__ = [REF]
    sys.modules["thunder"] = pkg
    ta, taf = types.ModuleType("torchaudio"), types.ModuleType("torchaudio.functional")

    def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
        assert norm == "slaney" and mel_scale == "slaney"
        return torch.from_numpy(slaney_mel_filterbank(n_freqs, n_mels, sample_rate, f_min, f_max)).t().contiguous()

    def mask_along_axis(*a, **k):
        raise RuntimeError("not used in eval")

    taf.melscale_fbanks, taf.mask_along_axis = melscale_fbanks, mask_along_axis
    ta.functional = taf
    sys.modules["torchaudio"], sys.modules["torchaudio.functional"] = ta, taf


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(os.environ.get("TS_GOLDEN_OUT", HERE), name)      # TS_GOLDEN_OUT: regenerate into a scratch directory and compare
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def check(a, b, atol, what):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    print(f"  oracle vs reference [{what}]: max abs err {err:.3e}")
    assert a.shape == b.shape and err <= atol, f"{what}: oracle != reference ({err} > {atol})"


def sd_numpy(sd, prefix=""):
    return {prefix + k.replace(".", "/"): v.detach().cpu().numpy() for k, v in sd.items()}


def rng_wave(seed, b, t, scale=0.1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal((b, t))).astype(np.float32))


def main():
    torch.set_num_threads(4)
    import_reference()
    from oracle import ctc as octc, decode as odec, frontend as ofe, primitives as oprim, tcs as otcs

    from thunder.blocks import conv1d_decoder, lengths_to_mask, linear_decoder, normalize_tensor
    from thunder.citrinet.blocks import CitrinetBlock, CitrinetEncoder, SqueezeExcite
    from thunder.ctc_loss import calculate_ctc
    from thunder.huggingface.transform import Wav2Vec2Preprocess
    from thunder.quartznet.blocks import MaskedConv1d, QuartznetBlock, QuartznetEncoder
    from thunder.quartznet.transform import FilterbankFeatures
    from thunder.text_processing.transform import BatchTextTransformer

    # ------------------------------------------------------------------ front end
    for tag, kw in (("qn", dict()), ("cn", dict(n_window_size=400, nfilt=80))):
        cfg = ofe.FrontendConfig(**kw)
        fb = FilterbankFeatures(**kw).eval()
        x = rng_wave(11, 3, 4000)
        x[1, 3200:] = 0.0          # asr_collate zero-pads (data/dataloader_utils.py:26-33)
        x[2, 2500:] = 0.0
        lengths = torch.tensor([4000.0, 3200.0, 2500.0])          # FLOAT lengths (A5)
        with torch.no_grad():
            pe = fb[0](x, lengths)[0]
            power, flen = fb[1](pe, lengths)
            lm = fb[2](power, flen)[0]
            feats, flen2 = fb[3](lm, flen)
            whole, wl = fb(x, lengths)
        assert torch.equal(feats, whole)
        st = ofe.filterbank_features(x, lengths, cfg, return_stages=True)
        check(st["preemph"], pe, 1e-7, f"{tag} preemph")
        check(st["power"], power, 2e-4 * float(power.max()), f"{tag} power")
        check(st["logmel"], lm, 2e-4, f"{tag} logmel")
        check(st["features"], feats, 1e-3, f"{tag} features")
        assert torch.equal(st["lengths"], flen)
        save(f"frontend_{tag}.npz", x=x, lengths=lengths, preemph=pe, power=power, logmel=lm, features=feats,
             feat_lengths=flen, mel_fb=fb[2].layer[0].fb[0])

    # ------------------------------------------------------------------ primitives
    xs = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((3, 4, 37)).astype(np.float32)) * 2 + 1
    lens = torch.tensor([37, 20, 5])
    mask = lengths_to_mask(lens, 37).unsqueeze(1)
    nm = normalize_tensor(xs, mask, div_guard=1e-5)
    nu = normalize_tensor(xs[:, 0], None, div_guard=1e-7)
    check(oprim.masked_normalize(xs, mask, 1e-5), nm, 1e-5, "masked normalize (A1)")
    check(oprim.unmasked_normalize(xs[:, 0], 1e-7), nu, 1e-5, "unmasked normalize")
    w2v_m = Wav2Vec2Preprocess(mask_input=True)(xs[:, 0], lens)[0]
    w2v_u = Wav2Vec2Preprocess(mask_input=False)(xs[:, 0], lens)[0]
    check(oprim.wav2vec2_preprocess(xs[:, 0], lens, True)[0], w2v_m, 1e-5, "w2v2 preprocess masked")
    check(oprim.wav2vec2_preprocess(xs[:, 0], lens, False)[0], w2v_u, 1e-5, "w2v2 preprocess unmasked")
    save("primitives.npz", x=xs, lengths=lens, masked=nm, unmasked=nu, w2v_masked=w2v_m, w2v_unmasked=w2v_u)

    # ------------------------------------------------------------------ blocks (small, eval + train)
    def load_into(module, sd):
        missing = module.load_state_dict(sd, strict=True)
        return missing

    block_cases = [
        # name, family, spec kwargs, T, lengths
        ("qn_res_k11", "quartznet", dict(in_ch=16, out_ch=32, repeat=3, kernel=11), 50, [50, 33, 7]),
        ("qn_stride2", "quartznet", dict(in_ch=16, out_ch=24, repeat=1, kernel=33, stride=2, residual=False), 61, [61, 40, 12]),
        ("qn_dil2", "quartznet", dict(in_ch=24, out_ch=24, repeat=1, kernel=13, dilation=2, residual=False), 45, [45, 30, 1]),
        ("qn_dense_k1", "quartznet", dict(in_ch=24, out_ch=48, repeat=1, kernel=1, residual=False, separable=False), 40, [40, 21, 3]),
        ("qn_stride2_rep2", "quartznet", dict(in_ch=16, out_ch=16, repeat=2, kernel=5, stride=2, residual=True), 64, [64, 37, 9]),
        ("cn_s1", "citrinet", dict(in_ch=16, out_ch=32, repeat=3, kernel=7, stride=1, family="citrinet"), 48, [48, 31, 6]),
        ("cn_s2", "citrinet", dict(in_ch=32, out_ch=32, repeat=2, kernel=9, stride=2, family="citrinet"), 51, [51, 30, 10]),
    ]
    blocks_out = {}
    for name, family, kw, T, lens in block_cases:
        spec = otcs.BlockSpec(**kw)
        sd = otcs.synth_encoder_state([spec], seed=zlib.crc32(name.encode()) % 1000 + 3)
        sd1 = {k[2:]: v for k, v in sd.items()}        # strip the "0." encoder index
        cls = QuartznetBlock if family == "quartznet" else CitrinetBlock
        mod = cls(spec.in_ch, spec.out_ch, repeat=spec.repeat, kernel_size=(spec.kernel,), stride=(spec.stride,),
                  dilation=(spec.dilation,), residual=spec.residual, separable=spec.separable)
        load_into(mod, sd1)
        x = torch.from_numpy(np.random.Generator(np.random.PCG64(17)).standard_normal((len(lens), spec.in_ch, T)).astype(np.float32))
        lengths = torch.tensor(lens)
        mod.eval()
        with torch.no_grad():
            y_eval, l_eval = mod(x, lengths)
        o_eval, ol = otcs.block_forward(spec, sd1, "", x, lengths)
        check(o_eval, y_eval, 1e-4, f"block {name} eval")
        assert torch.equal(ol, l_eval)
        mod.train()
        with torch.no_grad():
            y_train, _ = mod(x, lengths)
        stats = {}
        o_train, _ = otcs.block_forward(spec, sd1, "", x, lengths, training=True, new_stats=stats)
        check(o_train, y_train, 1e-4, f"block {name} train")
        new_sd = mod.state_dict()
        for k, v in stats.items():
            check(v, new_sd[k], 1e-5, f"block {name} {k}")
        blocks_out.update(sd_numpy(sd1, f"{name}/sd/"))
        blocks_out[f"{name}/x"] = x.numpy()
        blocks_out[f"{name}/lengths"] = lengths.numpy()
        blocks_out[f"{name}/y_eval"] = y_eval.numpy()
        blocks_out[f"{name}/y_train"] = y_train.numpy()
        blocks_out[f"{name}/out_lengths"] = l_eval.numpy()
        for k in stats:
            blocks_out[f"{name}/new/" + k.replace(".", "/")] = new_sd[k].numpy()
    save("blocks.npz", **blocks_out)

    # masked conv + SE on their own (ragged lengths, padded frames)  (A2, A3)
    torch.manual_seed(11)                      # the two modules below take torch's default init: seeded, so that the fixture regenerates bit for bit
    mc = MaskedConv1d(8, 8, 5, stride=1, padding=2, groups=8)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((2, 8, 20)).astype(np.float32))
    lengths = torch.tensor([20.0, 11.0])       # float lengths flow through get_seq_len (A5)
    with torch.no_grad():
        y, yl = mc(x, lengths)
    oy, oyl = otcs.masked_conv(x, lengths, mc.conv.weight.detach(), 1, 2, 1, 8)
    check(oy, y, 1e-6, "masked conv"); assert torch.equal(oyl, yl) and oyl.dtype == yl.dtype
    se = SqueezeExcite(16, 8).eval()
    xse = torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((2, 16, 13)).astype(np.float32))
    with torch.no_grad():
        yse = se(xse)
    check(otcs.squeeze_excite(xse, se.fc[0].weight.detach(), se.fc[2].weight.detach()), yse, 1e-6, "squeeze-excite")
    save("conv_se.npz", conv_w=mc.conv.weight, conv_x=x, conv_lengths=lengths, conv_y=y, conv_out_lengths=yl,
         se_w1=se.fc[0].weight, se_w2=se.fc[2].weight, se_x=xse, se_y=yse)

    # ------------------------------------------------------------------ whole QuartzNet5x5 + decoder (full-size weights from seed)
    arch = otcs.quartznet_arch(repeat_blocks=1)
    sd = otcs.synth_encoder_state(arch, seed=0, calibrate=True)
    dsd = otcs.synth_decoder_state(1024, 29, seed=1, gain=4.0)
    enc = QuartznetEncoder(repeat_blocks=1).eval()
    assert sorted(enc.state_dict().keys()) == sorted(sd.keys()), "state-dict key layout differs from the reference"
    enc.load_state_dict(sd, strict=True)
    dec = conv1d_decoder(1024, 29).eval()
    dec.load_state_dict(dsd, strict=True)
    fbank = FilterbankFeatures().eval()
    wav = rng_wave(1234, 2, 24000)
    wav[1, 17000:] = 0
    wl = torch.tensor([24000.0, 17000.0])
    with torch.no_grad():
        feats, fl = fbank(wav, wl)
        encd, el = enc(feats, fl)
        logits = dec(encd)
    ofeats, ofl = ofe.filterbank_features(wav, wl)
    oenc, oel = otcs.encoder_forward(arch, sd, ofeats, ofl)
    ologits = otcs.conv1d_decoder_forward(dsd, oenc)
    check(ologits, logits, 2e-3, "QN5x5 logits"); assert torch.equal(oel, el)
    n15 = sum(v.numel() for k, v in otcs.synth_encoder_state(otcs.quartznet_arch(repeat_blocks=3), seed=0).items()
              if "running" not in k and "num_batches" not in k) + 1024 * 29 + 29
    assert n15 == 18924381, n15          # SURVEY.md Appendix B probe
    e2e_labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    e2e_tt = BatchTextTransformer(tokens=list(e2e_labels))
    e2e_strings = e2e_tt.decode_prediction(logits.argmax(1))
    assert odec.decode_prediction(odec.argmax_classes(ologits.numpy()), odec.Vocab(list(e2e_labels))) == e2e_strings
    save("qn5x5_e2e.npz", strings=np.array(e2e_strings), wav_seed=1234, wav_shape=[2, 24000], wav_zero_from=[24000, 17000], wav_lengths=wl,
         enc_seed=0, dec_seed=1, logits=logits, out_lengths=el, enc_mean_abs=encd.abs().mean(),
         enc_sample=encd[:, ::64, ::10])

    # tiny citrinet encoder end to end
    carch = otcs.citrinet_arch(filters=[32, 32], kernel_sizes=[5, 7], strides=[2, 1], feat_in=16)
    # the reference hard-codes a 256-channel stem/640 head: citrinet/blocks.py:209-216,244-254
    csd = otcs.synth_encoder_state(carch, seed=7)
    cenc = CitrinetEncoder(filters=[32, 32], kernel_sizes=[5, 7], strides=[2, 1], feat_in=16).eval()
    assert sorted(cenc.state_dict().keys()) == sorted(csd.keys())
    cenc.load_state_dict(csd, strict=True)
    cx = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((2, 16, 41)).astype(np.float32))
    cl = torch.tensor([41, 23])
    with torch.no_grad():
        cy, cyl = cenc(cx, cl)
    coy, coyl = otcs.encoder_forward(carch, csd, cx, cl)
    check(coy, cy, 1e-4, "tiny citrinet encoder"); assert torch.equal(coyl, cyl)
    save("citrinet_tiny.npz", enc_seed=7, x=cx, lengths=cl, y_sample=cy[:, ::8, :], out_lengths=cyl,
         y_mean_abs=cy.abs().mean())

    # linear decoder
    ld = linear_decoder(32, 11, 0.1).eval()
    lsd = otcs.synth_decoder_state(32, 11, seed=2, linear=True)
    ld.load_state_dict(lsd, strict=True)
    lx = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((2, 32, 9)).astype(np.float32))
    with torch.no_grad():
        ly = ld(lx)
    check(otcs.linear_decoder_forward(lsd, lx), ly, 1e-5, "linear decoder")
    save("linear_decoder.npz", x=lx, y=ly, **sd_numpy(lsd, "sd/"))

    # ------------------------------------------------------------------ CTC
    rng = np.random.Generator(np.random.PCG64(21))
    B, V, T = 5, 29, 40
    logits = torch.from_numpy((2.0 * rng.standard_normal((B, V, T))).astype(np.float32)).requires_grad_(True)
    tlen = np.array([12, 7, 1, 25, 0])            # 25 > feasible for in_len 30 with repeats -> inf -> zeroed
    targets = np.full((B, 25), 28, dtype=np.int64)
    for b in range(B):
        targets[b, :tlen[b]] = rng.integers(0, 28, tlen[b])
    targets[3, :25] = 5                            # all-equal labels need 2S-1 frames: infeasible
    in_len = torch.tensor([40.0, 31.0, 40.0, 30.0, 17.0])
    loss = calculate_ctc(logits, torch.from_numpy(targets), in_len, torch.from_numpy(tlen), 28)
    loss.backward()
    oloss, ograd, onll = octc.calculate_ctc(logits.detach().numpy(), targets, in_len.numpy(), tlen, 28)
    check(np.array(oloss), loss.detach(), 1e-4, "ctc loss")
    check(ograd, logits.grad, 1e-5, "ctc grad")
    save("ctc.npz", logits=logits.detach(), targets=targets, input_lengths=in_len, target_lengths=tlen, blank=28,
         loss=loss.detach(), grad=logits.grad)

    # ------------------------------------------------------------------ greedy decode + encode
    labels = [" ", "a", "b", "c", "d", "e", "f", "g", "h", "i", "j", "k", "l", "m", "n", "o", "p", "q", "r", "s", "t",
              "u", "v", "w", "x", "y", "z", "'"]
    tt = BatchTextTransformer(tokens=list(labels))
    vocab = odec.Vocab(list(labels))
    assert vocab.itos == tt.vocab.itos and vocab.blank_idx == tt.vocab.blank_idx == 28
    pred = torch.from_numpy(rng.integers(0, 29, (4, 60)))
    pred[0, 10:30] = 28
    pred[1, :] = 28
    pred[2, 5:9] = 3
    strings = tt.decode_prediction(pred)
    assert odec.decode_prediction(pred.numpy(), vocab) == strings
    strings_norep = tt.decode_prediction(pred, remove_repeated=False)
    assert odec.decode_prediction(pred.numpy(), vocab, remove_repeated=False) == strings_norep
    texts = ["hello world", "it's a test", ""]
    enc_ids, enc_len = tt.encode(texts)
    oids, olen = odec.encode_chars(texts, vocab)
    assert np.array_equal(oids, enc_ids.numpy()) and np.array_equal(olen, enc_len.numpy())
    save("decode.npz", labels=np.array(labels), pred=pred, strings=np.array(strings), strings_norep=np.array(strings_norep),
         texts=np.array(texts), enc_ids=enc_ids, enc_len=enc_len)
    print("all fixtures written; oracle == reference on every case")


if __name__ == "__main__":
    main()
