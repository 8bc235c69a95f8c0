"""QuartzNet checkpoints (reference: src/thunder/quartznet/compatibility.py).

* `load_quartznet_checkpoint(path_or_enum)` imports a NeMo `.nemo` archive: tar extraction with the stdlib,
  YAML with PyYAML (omegaconf / torchaudio / wget are absent here), the reference's key renaming
  (compatibility.py:137-157) and a strict load into the same module tree.  Downloading is not possible offline:
  enum members resolve to `~/.thunder/<name>.nemo` and raise FileNotFoundError when the file is missing.
* `build_synthetic_quartznet` builds the same module with seeded random weights (no checkpoint needed) -- used by
  the tests and the benchmark, where the weights come from the oracle's platform-stable generator.
"""
from __future__ import annotations

import os
import tarfile
import tempfile
from enum import Enum
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from ..blocks import conv1d_decoder
from ..module import BaseCTCModule
from ..text_processing.transform import BatchTextTransformer
from .blocks import QuartznetEncoder
from .transform import FilterbankFeatures


class QuartznetCheckpoint(str, Enum):
    """Names of the NeMo checkpoints the reference knows (compatibility.py:45-58)."""
    QuartzNet5x5LS_En = "QuartzNet5x5LS-En"
    QuartzNet15x5Base_En = "QuartzNet15x5Base-En"
    QuartzNet15x5NR_En = "QuartzNet15x5NR-En"
    stt_ca_quartznet15x5 = "stt_ca_quartznet15x5"
    stt_de_quartznet15x5 = "stt_de_quartznet15x5"
    stt_es_quartznet15x5 = "stt_es_quartznet15x5"
    stt_fr_quartznet15x5 = "stt_fr_quartznet15x5"
    stt_it_quartznet15x5 = "stt_it_quartznet15x5"
    stt_pl_quartznet15x5 = "stt_pl_quartznet15x5"
    stt_ru_quartznet15x5 = "stt_ru_quartznet15x5"
    stt_zh_quartznet15x5 = "stt_zh_quartznet15x5"


ENGLISH_LABELS = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def load_components_from_quartznet_config(config: Union[str, Path, Dict]) -> Tuple[nn.Module, nn.Module, BatchTextTransformer]:
    """NeMo model_config.yaml -> (encoder, audio_transform, text_transform) (compatibility.py:71-124)."""
    import yaml
    if not isinstance(config, dict):
        with open(config, "r") as f:
            config = yaml.safe_load(f)
    enc_cfg = config["encoder"]
    body_cfg = enc_cfg["jasper"][1:-2]                       # first = stem, last two = dilated + 1x1 heads
    filters = [c["filters"] for c in body_cfg]
    kernel_sizes = [c["kernel"][0] for c in body_cfg]
    # consecutive identical (filters, kernel) entries are the repeated blocks
    uniq, repeat = [], 1
    for fk in zip(filters, kernel_sizes):
        if not uniq or uniq[-1] != fk:
            uniq.append(fk)
    repeat = max(1, len(body_cfg) // max(1, len(uniq)))
    encoder = QuartznetEncoder(feat_in=enc_cfg["feat_in"], filters=[u[0] for u in uniq],
                               kernel_sizes=[u[1] for u in uniq], repeat_blocks=repeat)
    pre = config["preprocessor"]
    sr = pre["sample_rate"]
    audio_transform = FilterbankFeatures(sample_rate=sr, n_window_size=int(pre["window_size"] * sr),
                                         n_window_stride=int(pre["window_stride"] * sr), n_fft=pre["n_fft"],
                                         preemph=0.97, nfilt=pre["features"], dither=pre.get("dither", 1e-5))
    text_transform = BatchTextTransformer(tokens=list(config["labels"]))
    return encoder, audio_transform, text_transform


def fix_encoder_name(key: str) -> str:
    """NeMo state-dict key -> this module tree (compatibility.py:137-144)."""
    return (key.replace("encoder.encoder.", "").replace("encoder.", "")
            .replace(".conv.weight", ".conv.weight").replace("mconv.", "mconv.")
            .replace(".res.0.", ".res.").replace("bn.", "layer.0."))


def load_quartznet_weights(encoder: nn.Module, decoder: nn.Module, weights_path: str):
    sd = torch.load(weights_path, map_location="cpu")
    enc = {}
    for k, v in sd.items():
        if not k.startswith("encoder."):
            continue
        nk = k[len("encoder.encoder."):] if k.startswith("encoder.encoder.") else k[len("encoder."):]
        # NeMo: "<blk>.mconv.<i>.conv.weight" (convs) / "<blk>.mconv.<i>.weight" (BN) / "<blk>.res.0.<j>..."
        parts = nk.split(".")
        if "res" in parts:
            r = parts.index("res")
            parts = parts[:r + 1] + parts[r + 2:]            # drop the dense-residual list index
        if parts[-2] != "conv" and parts[-1] in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            parts = parts[:-1] + ["layer", "0", parts[-1]]
        enc[".".join(parts)] = v
    encoder.load_state_dict(enc, strict=True)
    dec = {"weight": sd["decoder.decoder_layers.0.weight"], "bias": sd["decoder.decoder_layers.0.bias"]}
    decoder.load_state_dict(dec, strict=True)


def load_quartznet_checkpoint(checkpoint: Union[str, QuartznetCheckpoint], save_folder: Optional[str] = None,
                              augment_params: Optional[dict] = None) -> BaseCTCModule:
    """reference compatibility.py:161-201 (no download: the .nemo file must already be on disk)."""
    if isinstance(checkpoint, QuartznetCheckpoint):
        folder = Path(save_folder) if save_folder else Path.home() / ".thunder"
        nemo_path = folder / f"{checkpoint.value}.nemo"
    else:
        nemo_path = Path(checkpoint)
    if not nemo_path.exists():
        raise FileNotFoundError(f"{nemo_path} not found; this environment has no network access to download it")
    with tempfile.TemporaryDirectory() as tmp:
        with tarfile.open(nemo_path) as tar:
            tar.extractall(tmp)
        cfg = next(Path(tmp).rglob("model_config.yaml"))
        wts = next(Path(tmp).rglob("model_weights.ckpt"))
        encoder, audio_transform, text_transform = load_components_from_quartznet_config(cfg)
        decoder = conv1d_decoder(1024, text_transform.num_tokens)
        load_quartznet_weights(encoder, decoder, str(wts))
    return BaseCTCModule(encoder=encoder, decoder=decoder, audio_transform=audio_transform,
                         text_transform=text_transform, encoder_final_dimension=1024).eval()


def build_synthetic_quartznet(repeat_blocks: int = 3, labels: Optional[List[str]] = None,
                              encoder_state: Optional[Dict[str, torch.Tensor]] = None,
                              decoder_state: Optional[Dict[str, torch.Tensor]] = None) -> BaseCTCModule:
    """Reference-architecture QuartzNet (5x5: repeat_blocks=1, 15x5: 3) with caller-provided or default-initialised
    weights.  No pretrained weights are available offline."""
    text_transform = BatchTextTransformer(tokens=list(labels or ENGLISH_LABELS))
    encoder = QuartznetEncoder(repeat_blocks=repeat_blocks)
    decoder = conv1d_decoder(1024, text_transform.num_tokens)
    if encoder_state is not None:
        encoder.load_state_dict(encoder_state, strict=True)
    if decoder_state is not None:
        decoder.load_state_dict(decoder_state, strict=True)
    return BaseCTCModule(encoder=encoder, decoder=decoder, audio_transform=FilterbankFeatures(),
                         text_transform=text_transform, encoder_final_dimension=1024).eval()
