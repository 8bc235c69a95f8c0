"""Oracle CTC loss / greedy decode vs the reference (fixtures + the reference's own known answers)."""
import numpy as np
import torch

from oracle import ctc, decode

LABELS = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def test_ctc_value_and_gradient_match_reference(golden):
    g = golden("ctc.npz")
    loss, grad, nll = ctc.calculate_ctc(g["logits"], g["targets"], g["input_lengths"], g["target_lengths"], int(g["blank"]))
    np.testing.assert_allclose(loss, float(g["loss"]), atol=1e-4)
    np.testing.assert_allclose(grad, g["grad"], atol=1e-5)
    assert nll[3] == 0.0                       # infeasible alignment -> inf -> zeroed (A10)
    assert np.all(grad[3] == 0)
    assert np.all(grad[1][:, 31:] == 0)        # frames beyond the input length get no gradient


def test_ctc_matches_torch_on_random_cases():
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(3):
        B, V, T = 3, 7, 12
        logits = rng.standard_normal((B, V, T)).astype(np.float32)
        tl = rng.integers(1, 5, B)
        tg = rng.integers(0, V - 1, (B, 5))
        il = rng.integers(8, T + 1, B)
        lt = torch.from_numpy(logits).requires_grad_(True)
        lp = torch.log_softmax(lt.permute(2, 0, 1), dim=2)
        ref = torch.nn.functional.ctc_loss(lp, torch.from_numpy(tg), torch.from_numpy(il), torch.from_numpy(tl),
                                           blank=V - 1, reduction="mean", zero_infinity=True)
        ref.backward()
        loss, grad, _ = ctc.calculate_ctc(logits, tg, il, tl, V - 1)
        np.testing.assert_allclose(loss, float(ref.detach()), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(grad, lt.grad.numpy(), atol=5e-6)   # torch side is fp32


def test_greedy_decode_reference_known_answers():
    # reference: tests/text/test_transforms.py:59-91
    v = decode.Vocab(list(LABELS))
    blank = v.blank_idx
    assert decode.decode_prediction(np.full((1, 10), blank), v) == [""]
    a, b = v.stoi["a"], v.stoi["b"]
    assert decode.decode_prediction(np.array([[a] * 5 + [b] * 5]), v) == ["ab"]
    assert decode.decode_prediction(np.array([[a] * 4 + [blank] + [a] * 4]), v) == ["aa"]


def test_encode_reference_known_answer():
    # reference: tests/text/test_transforms.py:41-56 -- start/end tokens appended after blank/pad/unk
    labels = [" "] + [chr(ord("a") + i) for i in range(26)]
    v = decode.Vocab(labels, blank_token="<blank>", pad_token="<pad>", unknown_token="<unk>",
                     start_token="<bos>", end_token="<eos>")
    ids, lens = decode.encode_chars(["hello world"], v)
    assert ids[0].tolist() == [v.stoi["<bos>"], 8, 5, 12, 12, 15, 0, 23, 15, 18, 12, 4, v.stoi["<eos>"]]
    assert lens.tolist() == [13]


def test_decode_and_encode_fixture(golden):
    g = golden("decode.npz")
    v = decode.Vocab([str(s) for s in g["labels"]])
    assert decode.decode_prediction(g["pred"], v) == [str(s) for s in g["strings"]]
    assert decode.decode_prediction(g["pred"], v, remove_repeated=False) == [str(s) for s in g["strings_norep"]]
    ids, lens = decode.encode_chars([str(s) for s in g["texts"]], v)
    assert np.array_equal(ids, g["enc_ids"]) and np.array_equal(lens, g["enc_len"])
