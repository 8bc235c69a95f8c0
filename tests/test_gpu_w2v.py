"""GPU parity: wav2vec2 waveform normalisation (ts_w2v_preprocess) vs the reference fixture and the oracle."""
import numpy as np
import pytest
import torch

from oracle import primitives as oprim

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mask_input", [False, True])
def test_matches_reference_fixture(golden, mask_input):
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    g = golden("primitives.npz")
    x = torch.from_numpy(g["x"][:, 0])
    lengths = torch.from_numpy(g["lengths"])
    y, yl = Wav2Vec2Preprocess(mask_input=mask_input)(x.cuda(), lengths.cuda())
    assert torch.equal(yl.cpu(), lengths)
    want = g["w2v_masked"] if mask_input else g["w2v_unmasked"]
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=2e-5)      # the reference's own tolerance is 1e-3


@pytest.mark.parametrize("mask_input", [False, True])
def test_c5_sized_clips_match_oracle(mask_input):
    """16 x 20 s clips (config C5), ragged float lengths, a DC offset so that the masked-variance quirk matters."""
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    g = torch.Generator().manual_seed(3)
    x = 0.1 * torch.randn(16, 320000, generator=g) + 0.03
    lengths = torch.floor(torch.linspace(0.5, 1.0, 16) * 320000)
    y, _ = Wav2Vec2Preprocess(mask_input=mask_input)(x.cuda(), lengths.cuda())
    ref, _ = oprim.wav2vec2_preprocess(x.double(), lengths, mask_input)
    assert float((y.cpu().double() - ref).abs().max()) <= 2e-4


def test_cpu_tensors_fail_loudly():
    from thunder_speech_amd.huggingface.transform import Wav2Vec2Preprocess
    with pytest.raises(RuntimeError):
        Wav2Vec2Preprocess()(torch.zeros(2, 100), torch.tensor([100, 50]))
