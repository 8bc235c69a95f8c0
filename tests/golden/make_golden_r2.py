"""Round-2 golden fixtures, generated from the REAL reference modules in the build container (needs /root/reference):

  r2_misc.npz
    * SpecCutout / SpecAugment mask geometry under torch.manual_seed: the reference's own `_create_mask`
      (quartznet/spec_augment.py:60-75) and SpecAugment.forward driven through a stand-in `mask_along_axis` that restates
      torchaudio 0.12.0 (torchaudio is absent from this image) -- inputs, seeds, masked outputs;
    * asr_collate (data/dataloader_utils.py:17-33) on ragged clips with a tie in lengths;
    * BatchTextTransformer.encode / decode_prediction known answers incl. start / end tokens and an unknown token;
    * block fixtures in TRAIN mode for the configurations round 1 raised on: a Citrinet block with squeeze-excite and a
      strided last repeat + strided residual, a QuartzNet block with a strided residual -- forward, running statistics
      and input gradient from the reference's autograd (dropout 0: the mask of nn.Dropout is generator-dependent).

Every output is compared with the oracle restatement while generating (abort on mismatch).

    python tests/golden/make_golden_r2.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import check, import_reference, save, sd_numpy  # noqa: E402


def torchaudio_mask_along_axis(specgram, mask_param, mask_value, axis):
    """torchaudio.functional.mask_along_axis, torchaudio 0.12.0 (published form), for the stand-in module."""
    if axis not in [1, 2]:
        raise ValueError("Only Frequency and Time masking are supported")
    shape = specgram.size()
    specgram = specgram.reshape([-1] + list(shape[-2:]))
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (specgram.size(axis) - value)
    mask_start = (min_value.long()).squeeze()
    mask_end = (min_value.long() + value.long()).squeeze()
    mask = torch.arange(0, specgram.shape[axis], device=specgram.device, dtype=specgram.dtype)
    mask = (mask >= mask_start) & (mask < mask_end)
    if axis == 1:
        mask = mask.unsqueeze(-1)
    assert mask_end - mask_start < mask_param
    specgram = specgram.masked_fill(mask, mask_value)
    return specgram.reshape(shape[:-2] + specgram.shape[-2:])


def main():
    torch.set_num_threads(4)
    import_reference()
    sys.modules["torchaudio.functional"].mask_along_axis = torchaudio_mask_along_axis
    from oracle import augment as oaug, dataprep as odp, tcs as otcs

    from thunder.citrinet.blocks import CitrinetBlock
    from thunder.data.dataloader_utils import asr_collate
    from thunder.quartznet.blocks import QuartznetBlock
    from thunder.quartznet.spec_augment import SpecAugment, SpecCutout
    from thunder.text_processing.transform import BatchTextTransformer

    out = {}
    rng = np.random.Generator(np.random.PCG64(21))
    # ------------------------------------------------------------------ SpecAugment / SpecCutout
    x = torch.from_numpy(rng.standard_normal((2, 64, 301)).astype(np.float32))
    out["spec_x"] = x
    for i, (seed, kw) in enumerate([(3, dict(freq_masks=2, time_masks=2, freq_width=15, time_width=25)),
                                    (4, dict(freq_masks=0, time_masks=3, freq_width=10, time_width=50)),
                                    (5, dict(freq_masks=1, time_masks=0, freq_width=20, time_width=10))]):
        mod = SpecAugment(**kw).train()
        torch.manual_seed(seed)
        y = mod(x)
        torch.manual_seed(seed)
        table = oaug.draw_table(oaug.torch_rand2, 64, 301, n_time=kw["time_masks"], time_width=kw["time_width"],
                                n_freq=kw["freq_masks"], freq_width=kw["freq_width"])
        check(oaug.apply_table(x, table), y, 0.0, f"SpecAugment {kw}")
        assert torch.equal(mod.eval()(x), x)
        out[f"specaug{i}_seed"], out[f"specaug{i}_cfg"] = seed, np.asarray([kw["time_masks"], kw["time_width"], kw["freq_masks"], kw["freq_width"]])
        out[f"specaug{i}_table"] = table           # y == x with the table's rectangles zeroed (asserted exactly above)
        if i == 0:
            out["specaug0_y"] = y
    for i, (seed, kw) in enumerate([(7, dict(rect_masks=3, time_width=30, freq_width=20)), (8, dict(rect_masks=5, time_width=5, freq_width=40))]):
        mod = SpecCutout(**kw).train()
        torch.manual_seed(seed)
        y = mod(x)
        torch.manual_seed(seed)
        table = oaug.draw_table(oaug.torch_rand2, 64, 301, n_cutout=kw["rect_masks"], cut_time_width=kw["time_width"],
                                cut_freq_width=kw["freq_width"])
        check(oaug.apply_table(x, table), y, 0.0, f"SpecCutout {kw}")
        out[f"cutout{i}_seed"], out[f"cutout{i}_cfg"] = seed, np.asarray([kw["rect_masks"], kw["time_width"], kw["freq_width"]])
        out[f"cutout{i}_table"] = table
        if i == 0:
            out["cutout0_y"] = y

    # ------------------------------------------------------------------ asr_collate
    lens = [700, 1000, 700, 333, 2]
    clips = [torch.from_numpy(rng.standard_normal((1, n)).astype(np.float32)) for n in lens]
    samples = [(c, f"text {i}") for i, c in enumerate(clips)]
    pa, pl, texts = asr_collate(samples)
    oa, ol, ot = odp.asr_collate(samples)
    check(oa, pa, 0.0, "asr_collate audio")
    check(ol, pl, 0.0, "asr_collate lengths")
    assert ot == texts and pl.dtype == torch.float32
    for i, c in enumerate(clips):
        out[f"collate_clip{i}"] = c
    out["collate_audio"], out["collate_lengths"] = pa, pl
    out["collate_order"] = np.asarray([int(t.split()[1]) for t in texts])

    # ------------------------------------------------------------------ text encode / decode
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    texts_in = ["hello world", "it's", "", "zebra q"]
    t1 = BatchTextTransformer(tokens=labels)
    e1, l1 = t1.encode(texts_in)
    t2 = BatchTextTransformer(tokens=labels, start_token="<bos>", end_token="<eos>", unknown_token="<unk>")
    texts_unk = ["héllo wörld", "ok", "x"]
    e2, l2 = t2.encode(texts_unk)
    e3, l3 = t1.encode(texts_unk)                      # no unknown token: out-of-vocabulary characters are dropped
    out.update(enc1=e1, len1=l1, enc2=e2, len2=l2, enc3=e3, len3=l3)
    out["enc_texts"] = np.asarray(texts_in)
    out["enc_texts_unk"] = np.asarray(texts_unk)
    out["dec1"] = np.asarray(t1.decode_prediction(e1, remove_repeated=False))

    # ------------------------------------------------------------------ training-mode blocks round 1 raised on
    def run_block(name, blk, arch_kw, x, lengths):
        blk.train()
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        xg = x.clone().requires_grad_(True)
        y, yl = blk(xg, lengths)
        w = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * w).sum().backward()
        out.update(sd_numpy(sd0, f"{name}/sd/"))
        out.update(sd_numpy(blk.state_dict(), f"{name}/sd_after/"))
        out.update({f"{name}/x": x, f"{name}/lengths": lengths, f"{name}/y": y.detach(), f"{name}/out_lengths": yl, f"{name}/w": w,
                    f"{name}/dx": xg.grad})
        for k, p in blk.named_parameters():
            out[f"{name}/grad/" + k.replace(".", "/")] = p.grad.detach()
        return y

    def perturb_bn(blk, seed):
        g = torch.Generator().manual_seed(seed)
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.data = 1.0 + 0.2 * torch.randn(m.num_features, generator=g)
                m.bias.data = 0.1 * torch.randn(m.num_features, generator=g)

    torch.manual_seed(31)
    cb = CitrinetBlock(16, 24, repeat=2, kernel_size=(5,), stride=(2,), separable=True, dropout=0.0)
    perturb_bn(cb, 1)
    xc = torch.from_numpy(rng.standard_normal((3, 16, 41)).astype(np.float32))
    lc = torch.tensor([41, 30, 17])
    run_block("cn_train_s2", cb, None, xc, lc)
    torch.manual_seed(32)
    qb = QuartznetBlock(16, 24, repeat=2, kernel_size=(5,), stride=(2,), separable=True, dropout=0.0)
    perturb_bn(qb, 2)
    run_block("qn_train_s2", qb, None, xc, lc)
    save("r2_misc.npz", **out)

    # ------------------------------------------------------------------ margin-calibrated end-to-end fixture (QuartzNet5x5)
    # Random decoder weights give tie-rich logits (round 1 had to mask "undecided" frames out).  Here the decoder is FITTED
    # (ridge regression on the reference encoder's output) so that every frame has a clear winner: the fp32 reference logits
    # are ~10 x one-hot of a label sequence with repeats and blanks, and greedy strings can be compared for EQUALITY.
    from oracle import decode as odec, frontend as ofe
    from thunder.blocks import conv1d_decoder
    from thunder.quartznet.blocks import QuartznetEncoder
    from thunder.quartznet.transform import FilterbankFeatures
    arch = otcs.quartznet_arch(repeat_blocks=1)
    sd = otcs.synth_encoder_state(arch, seed=0, calibrate=True)
    enc = QuartznetEncoder(repeat_blocks=1).eval()
    enc.load_state_dict(sd, strict=True)
    fbank = FilterbankFeatures().eval()
    wrng = np.random.Generator(np.random.PCG64(4321))
    wav = torch.from_numpy((0.1 * wrng.standard_normal((2, 24000))).astype(np.float32))
    wav[1, 17000:] = 0
    wl = torch.tensor([24000.0, 17000.0])
    with torch.no_grad():
        feats, fl = fbank(wav, wl)
        encd, el = enc(feats, fl)
    b, c, t = encd.shape
    labels = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    text = "hello world it's a test of the margin "
    ids = [labels.index(ch) for ch in text]
    lab = torch.full((b, t), 28, dtype=torch.long)                 # frames beyond the length hold one constant vector: blank
    for bi in range(b):
        seq, i = [], bi * 7
        while len(seq) < int(el[bi]):
            seq += [ids[i % len(ids)]] * 2 + [28]                   # every character held 2 frames, then a blank
            i += 1
        lab[bi, : int(el[bi])] = torch.tensor(seq[: int(el[bi])])
    X = encd.permute(0, 2, 1).reshape(-1, c).double()
    Y = torch.zeros(b * t, 29, dtype=torch.float64)
    Y[torch.arange(b * t), lab.reshape(-1)] = 10.0
    K = X @ X.T
    lam = 0.1 * float(torch.trace(K)) / (b * t)
    W = (X.T @ torch.linalg.solve(K + lam * torch.eye(b * t, dtype=torch.float64), Y)).T.float()      # [29, 1024]
    dec = conv1d_decoder(1024, 29).eval()
    dsd = {"weight": W.unsqueeze(-1).contiguous(), "bias": torch.zeros(29)}
    dec.load_state_dict(dsd, strict=True)
    with torch.no_grad():
        logits = dec(encd)
    top2 = torch.sort(logits, dim=1).values[:, -2:, :]
    margin = float((top2[:, 1] - top2[:, 0]).min())
    assert torch.equal(logits.argmax(1), lab) and margin > 1.5, margin
    tt = BatchTextTransformer(tokens=list(labels))
    strings = tt.decode_prediction(logits.argmax(1))
    ofeats, ofl = ofe.filterbank_features(wav, wl)
    oenc, oel = otcs.encoder_forward(arch, sd, ofeats, ofl)
    ologits = otcs.conv1d_decoder_forward(dsd, oenc)
    check(ologits, logits, 2e-3, "QN5x5 margin-calibrated logits")
    assert odec.decode_prediction(odec.argmax_classes(ologits.numpy()), odec.Vocab(list(labels))) == strings
    print("  margin-calibrated fixture: min top-1/top-2 margin", margin, "strings", strings)
    save("qn5x5_e2e_margin.npz", strings=np.array(strings), wav_seed=4321, wav_shape=[2, 24000], wav_zero_from=[24000, 17000],
         wav_lengths=wl, enc_seed=0, dec_weight=dsd["weight"], dec_bias=dsd["bias"], logits=logits, labels=lab, out_lengths=el,
         min_margin=margin)


if __name__ == "__main__":
    main()
