"""CPU-only checks of the boundary: the shared library exports every symbol the header declares, the ctypes
structs match the header, the host-side packers implement the documented fragment layouts, and the module
surface mirrors the reference (state-dict keys, constructor errors).  No kernel is launched here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "thunder_speech_amd.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from thunder_speech_amd import build as b
    from thunder_speech_amd._lib import EXPORTED_SYMBOLS
    path = b.build(verbose=False)              # hipcc cross-compiles gfx950 without a GPU
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    defined = {line.split()[-1] for line in nm.splitlines() if line.strip()}
    declared = _declared_functions()
    assert declared and set(declared) == set(EXPORTED_SYMBOLS)
    assert not [s for s in declared if s not in defined]
    # the code object really targets gfx950
    out = subprocess.run(["strings", "-n", "6", path], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_library_loads_and_answers_version_queries():
    from thunder_speech_amd import _lib
    L = _lib.lib()
    assert L.ts_abi_version() == _lib.ABI_VERSION == 11
    assert L.ts_build_target() == b"gfx950"
    for t in (1, 127, 128, 129, 751, 1501, 2001):
        assert L.ts_time_pitch(t) == _lib.time_pitch(t) and _lib.time_pitch(t) % 128 == 0 and _lib.time_pitch(t) >= t


def test_ctypes_structs_match_header_field_order():
    from thunder_speech_amd import _lib
    src = open(HEADER).read()
    for struct, cls in (("ts_tcs_desc", _lib.TcsDesc), ("ts_frontend_desc", _lib.FrontendDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([a-z_0-9]+)\s*$", part.strip())[0])
        assert names == [f[0] for f in cls._fields_], struct


def test_pack_pw_frags_layout():
    from thunder_speech_amd import plan
    w = torch.arange(40 * 70, dtype=torch.float32).reshape(40, 70) / 64.0
    fr = plan.pack_pw_frags(w).float()
    assert fr.shape == (2, 8, 64, 8)          # cout 40 -> 64 (2 tiles), cin 70 -> 128 (8 k-steps)
    for (cot, ks, lane, j) in [(0, 0, 0, 0), (0, 3, 37, 5), (1, 4, 7, 2), (1, 7, 63, 7), (0, 4, 33, 6)]:
        n, h = lane & 31, lane >> 5
        co, ci = cot * 32 + n, ks * 16 + 8 * h + j
        want = float(w[co, ci].to(torch.bfloat16)) if co < 40 and ci < 70 else 0.0
        assert float(fr[cot, ks, lane, j]) == want


@pytest.mark.parametrize("k,stride,dil", [(33, 1, 1), (39, 1, 1), (75, 1, 1), (33, 2, 1), (87, 1, 2), (5, 1, 1), (13, 1, 3)])
def test_pack_dw_taps_is_the_toeplitz_of_the_conv(k, stride, dil):
    """out[t + i] = sum_v row_i[v] * x[t*stride - padl4 + v] must equal the reference depthwise conv."""
    from thunder_speech_amd import plan
    from thunder_speech_amd.blocks import get_same_padding
    pad = get_same_padding(k, stride, dil)
    g = torch.Generator().manual_seed(k)
    w = torch.randn(3, 1, k, generator=g).to(torch.bfloat16).float()
    taps, nk = plan.pack_dw_taps(w, stride, dil, pad)
    assert taps.shape == (64, 4, 4 * nk) and nk % plan.NKP == 0 and nk == plan.dw_ksteps(k, stride, dil, pad)
    x = torch.randn(1, 3, 200, generator=g)
    ref = torch.nn.functional.conv1d(x, w, None, stride, pad, dil, groups=3)[0]
    padl4 = plan.round_up(pad, 4)
    xp = torch.nn.functional.pad(x[0], (padl4, 4 * nk + 8))
    for t in (0, 4, 8, 40):
        for i in range(4):
            if t + i >= ref.shape[1]:
                continue
            win = xp[:, t * stride: t * stride + 4 * nk]
            got = (taps[:3, i].float() * win).sum(-1)
            np.testing.assert_allclose(got.numpy(), ref[:, t + i].numpy(), atol=1e-4)
    assert torch.all(taps[3:] == 0)


def test_tap_fragments_layout():
    from thunder_speech_amd import plan
    taps = torch.arange(128 * 4 * 12, dtype=torch.float32).reshape(128, 4, 12)
    fr = plan.tap_fragments(taps)
    assert fr.shape == (2, 4, 3, 64, 4)
    for (chunk, wave, k, lane) in [(0, 0, 0, 0), (1, 3, 2, 63), (0, 2, 1, 37), (1, 0, 0, 5)]:
        ch, row = chunk * 64 + wave * 16 + lane // 4, lane % 4
        assert torch.equal(fr[chunk, wave, k, lane], taps[ch, row, 4 * k: 4 * k + 4])


def test_state_dict_keys_match_reference_layout():
    """Same keys/shapes as the reference module tree (pinned through the oracle's synthetic state dict, which the
    golden generator loaded strict=True into the real reference QuartznetEncoder)."""
    from oracle import tcs as otcs
    from thunder_speech_amd.quartznet.blocks import QuartznetEncoder
    from thunder_speech_amd.blocks import conv1d_decoder, linear_decoder
    for rb, n in ((1, 225), (3, 635)):
        enc = QuartznetEncoder(repeat_blocks=rb)
        ours = enc.state_dict()
        ref = otcs.synth_encoder_state(otcs.quartznet_arch(repeat_blocks=rb), seed=0)
        assert len(ours) == n and set(ours) == set(ref)
        assert all(tuple(ours[k].shape) == tuple(ref[k].shape) for k in ref)
        enc.load_state_dict(ref, strict=True)
    assert sum(p.numel() for p in QuartznetEncoder(repeat_blocks=3).parameters()) + 1024 * 29 + 29 == 18924381
    assert set(conv1d_decoder(1024, 29).state_dict()) == {"weight", "bias"}
    assert set(linear_decoder(32, 11, 0.1).state_dict()) == {"2.weight", "2.bias"}


def test_frontend_module_surface():
    from thunder_speech_amd.quartznet.transform import FilterbankFeatures, patch_stft
    fb = FilterbankFeatures()
    assert list(fb.state_dict()) == ["1.window", "2.layer.0.fb"]
    assert fb[1].n_fft == 512 and fb[1].hop_length == 160 and fb[1].win_length == 320 and fb[1].stft_func is torch.stft
    assert fb[1].get_sequence_length(torch.tensor([1234.0, 159.0])).tolist() == [8, 1]
    patch_stft(fb)
    with pytest.raises(ValueError):
        FilterbankFeatures(num_cutout_masks=1, num_time_masks=1)
    with pytest.raises(ValueError):
        FilterbankFeatures(n_window_size=0)


def test_error_conventions():
    from thunder_speech_amd.blocks import get_same_padding
    from thunder_speech_amd.finetune import FinetuneCTCModule
    from thunder_speech_amd.quartznet.blocks import InitMode, QuartznetBlock, init_weights
    from thunder_speech_amd.registry import load_pretrained
    with pytest.raises(ValueError):
        get_same_padding(3, 2, 2)
    with pytest.raises(ValueError):
        QuartznetBlock(8, 8, kernel_size=(3,), stride=(2,), dilation=(2,))
    with pytest.raises(ValueError):
        init_weights(torch.nn.Conv1d(2, 2, 1), "bogus")
    init_weights(torch.nn.Conv1d(2, 2, 1), InitMode.kaiming_normal)
    with pytest.raises(KeyError):
        load_pretrained("no_such_checkpoint")
    with pytest.raises(ValueError):
        FinetuneCTCModule("QuartzNet5x5_synthetic", tokens=["a", "b"])
    with pytest.raises(ValueError):
        FinetuneCTCModule("QuartzNet5x5_synthetic", decoder_class=lambda c, n: torch.nn.Conv1d(c, n, 1))
    m = FinetuneCTCModule("QuartzNet5x5_synthetic", decoder_class=lambda c, n: torch.nn.Conv1d(c, n, 1), tokens=["a", "b"])
    assert m.decoder.out_channels == 3 and m.encoder_final_dimension == 1024


def test_compute_paths_refuse_cpu_tensors():
    from thunder_speech_amd.registry import load_pretrained
    m = load_pretrained("QuartzNet5x5_synthetic")
    assert not m.training
    with pytest.raises(RuntimeError, match="no CPU"):
        m.predict(torch.zeros(1, 16000))
    with pytest.raises(RuntimeError, match="no CPU"):
        m.encoder(torch.zeros(1, 64, 50), torch.tensor([50]))


def test_text_transform_known_answers():
    # reference: tests/text/test_transforms.py:41-91
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    labels = [" "] + [chr(ord("a") + i) for i in range(26)]
    tt = BatchTextTransformer(labels, blank_token="<blank>", pad_token="<pad>", unknown_token="<unk>",
                              start_token="<bos>", end_token="<eos>")
    ids, lens = tt.encode(["hello world"])
    v = tt.vocab
    assert ids[0].tolist() == [v.stoi["<bos>"], 8, 5, 12, 12, 15, 0, 23, 15, 18, 12, 4, v.stoi["<eos>"]]
    tt2 = BatchTextTransformer(labels + ["'"])
    blank, a, b = tt2.vocab.blank_idx, tt2.vocab.stoi["a"], tt2.vocab.stoi["b"]
    assert tt2.decode_prediction(torch.full((1, 10), blank)) == [""]
    assert tt2.decode_prediction(torch.tensor([[a] * 5 + [b] * 5])) == ["ab"]
    assert tt2.decode_prediction(torch.tensor([[a] * 4 + [blank] + [a] * 4])) == ["aa"]
    assert tt2.num_tokens == 29 and blank == 28


def test_decode_fixture_matches_reference(golden):
    from thunder_speech_amd.text_processing.transform import BatchTextTransformer
    g = golden("decode.npz")
    tt = BatchTextTransformer([str(s) for s in g["labels"]])
    pred = torch.from_numpy(g["pred"])
    assert tt.decode_prediction(pred) == [str(s) for s in g["strings"]]
    assert tt.decode_prediction(pred, remove_repeated=False) == [str(s) for s in g["strings_norep"]]
    ids, lens = tt.encode([str(s) for s in g["texts"]])
    assert np.array_equal(ids.numpy(), g["enc_ids"]) and np.array_equal(lens.numpy(), g["enc_len"])


_ASAN_DRIVER = r'''
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
assert L.ts_abi_version() == 11
L.ts_build_target.restype = C.c_char_p
assert L.ts_build_target() == b"gfx950"
assert [L.ts_time_pitch(t) for t in (1, 128, 751, 1501)] == [512, 512, 1152, 1920]
# argument validation returns before any launch: null pointers, non-positive sizes, misaligned pitches (TS_EINVAL = -1, TS_EUNSUPPORTED = -2)
class Desc(C.Structure):
    _fields_ = [(n, i32) for n in ("batch", "c_in", "c_out", "t_in", "t_out", "pitch_in", "pitch_out", "kernel", "stride", "dilation", "padding", "depthwise",
                                   "relu", "out_fp32", "c_res", "pitch_res", "t_res", "res_stride", "dw_ksteps", "flags")] + \
               [(n, vp) for n in ("dw_taps", "dw_taps_raw", "pw_w", "res_w", "pw_w16", "res_w16", "bias", "se_y", "se_gate", "stats")]
d = Desc()
L.ts_tcs_subblock_fwd.argtypes = [C.POINTER(Desc), vp, vp, vp, vp, vp, vp]
assert L.ts_tcs_subblock_fwd(None, None, None, None, None, None, None) == -1
buf = (C.c_char * 4096)()
d.batch, d.c_in, d.c_out, d.t_out, d.pitch_in, d.pitch_out, d.kernel, d.stride, d.dilation = 1, 64, 64, 100, 513, 512, 1, 1, 1
d.pw_w = d.bias = C.addressof(buf)
assert L.ts_tcs_subblock_fwd(C.byref(d), buf, buf, None, None, buf, None) == -1              # pitch_in % 8
d.pitch_in, d.stride, d.dilation = 512, 2, 2
assert L.ts_tcs_subblock_fwd(C.byref(d), buf, buf, None, None, buf, None) == -1              # stride AND dilation
d.stride, d.dilation, d.kernel = 1, 1, 5
assert L.ts_tcs_subblock_fwd(C.byref(d), buf, buf, None, None, buf, None) == -2              # dense K > 1
L.ts_tcs_pointwise_tile_frames.argtypes = [i32, i32, i32]
assert L.ts_tcs_pointwise_tile_frames(32, 512, 501) == 64 and L.ts_tcs_pointwise_tile_frames(0, 512, 501) == -1
L.ts_greedy_decode.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
assert L.ts_greedy_decode(None, 1, 1, 1, 1, None, None, None, None) == -1
assert L.ts_greedy_decode(buf, 2, 29, 100, 64, buf, buf, buf, None) == -1                     # pitch < frames
L.ts_train_set_deterministic.argtypes = [vp, i64]
assert L.ts_train_set_deterministic(buf, 0) == -1 and L.ts_train_set_deterministic(None, 0) == 0
L.ts_train_pwconv_wgrad_workspace.argtypes = [i32, i32, i32]
L.ts_train_pwconv_wgrad_workspace.restype = i64
assert L.ts_train_pwconv_wgrad_workspace(32, 512, 512) >= 512 * 512
L.ts_train_pwconv_wgrad_multi_parts.argtypes = [i32, i32, i32]
assert 1 <= L.ts_train_pwconv_wgrad_multi_parts(32, 512, 512) <= 32
L.ts_train_dwconv_bwd.argtypes = [vp] * 7 + [i32] * 11 + [vp]
assert L.ts_train_dwconv_bwd(None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 8, 8, 0, None) == -1
L.ts_ctc_workspace_bytes.restype = i64
L.ts_ctc_workspace_bytes.argtypes = [i32, i32, i32, i32]
assert L.ts_ctc_workspace_bytes(32, 501, 160, 29) > 0
print("ASAN-DRIVER-OK")
'''


def test_host_code_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """`python -m thunder_speech_amd.build --asan`: the extern "C" launchers compiled host-only with -fsanitize=address,undefined and driven through
    the paths that return before a launch (version queries, argument validation, workspace arithmetic).  Any sanitizer report aborts the child."""
    from thunder_speech_amd import build as b
    path = b.build_asan(verbose=False)
    driver = tmp_path / "asan_driver.py"
    driver.write_text(_ASAN_DRIVER)
    env = dict(os.environ, LD_PRELOAD=b.asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    import sys
    r = subprocess.run([sys.executable, str(driver), path], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ASAN-DRIVER-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
