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
from ..utils import BaseCheckpoint
from .blocks import QuartznetEncoder
from .transform import FilterbankFeatures


class QuartznetCheckpoint(BaseCheckpoint):
    """Names of the NeMo checkpoints the reference knows (compatibility.py:45-58)."""
    QuartzNet5x5LS_En = "QuartzNet5x5LS-En"
    QuartzNet15x5Base_En = "QuartzNet15x5Base-En"
    QuartzNet15x5Base_Zh = "QuartzNet15x5Base-Zh"
    QuartzNet15x5NR_En = "QuartzNet15x5NR-En"
    stt_ca_quartznet15x5 = "stt_ca_quartznet15x5"
    stt_de_quartznet15x5 = "stt_de_quartznet15x5"
    stt_es_quartznet15x5 = "stt_es_quartznet15x5"
    stt_fr_quartznet15x5 = "stt_fr_quartznet15x5"
    stt_it_quartznet15x5 = "stt_it_quartznet15x5"
    stt_pl_quartznet15x5 = "stt_pl_quartznet15x5"
    stt_ru_quartznet15x5 = "stt_ru_quartznet15x5"
    stt_en_quartznet15x5 = "stt_en_quartznet15x5"
    stt_zh_quartznet15x5 = "stt_zh_quartznet15x5"


ENGLISH_LABELS = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def _section(conf: Dict, name: str) -> Dict:
    """NeMo wrote `encoder: {params: {...}}` in the QuartzNet-era configs and `encoder: {...}` later; accept both."""
    sec = conf[name]
    return sec["params"] if isinstance(sec, dict) and "params" in sec else sec


def load_components_from_quartznet_config(config: Union[str, Path, Dict], augment_params: Optional[dict] = None
                                          ) -> Tuple[nn.Module, nn.Module, BatchTextTransformer]:
    """NeMo model_config.yaml -> (encoder, audio_transform, text_transform) (compatibility.py:71-124): the body blocks are
    the `jasper` entries between the stem and the two heads, one QuartznetBlock per entry."""
    import yaml
    params = dict(augment_params or {})
    if not isinstance(config, dict):
        with open(config, "r") as f:
            config = yaml.safe_load(f)
    body_cfg = _section(config, "encoder")["jasper"][1:-2]
    # the reference passes one (filters, kernel) pair per body entry with repeat_blocks = 1: a 15x5 config lists its
    # 15 blocks explicitly, which gives the same module tree / state-dict keys as filters x 5 with repeat_blocks = 3
    encoder = QuartznetEncoder(filters=[c["filters"] for c in body_cfg], kernel_sizes=[c["kernel"][0] for c in body_cfg],
                               dropout=params.pop("dropout", 0.0))
    pre = _section(config, "preprocessor")
    sr = pre["sample_rate"]
    audio_transform = FilterbankFeatures(sample_rate=sr, n_window_size=int(pre["window_size"] * sr),
                                         n_window_stride=int(pre["window_stride"] * sr), n_fft=pre["n_fft"],
                                         nfilt=pre["features"], dither=pre["dither"], **params)
    labels = config["labels"] if "labels" in config else _section(config, "decoder")["vocabulary"]
    text_transform = BatchTextTransformer(tokens=list(labels))
    return encoder, audio_transform, text_transform


def fix_encoder_name(key: str) -> str:
    """NeMo state-dict key -> this module tree (compatibility.py:137-144): drop the `encoder.` prefixes and the
    dense-residual list index, and put BatchNorm entries behind the `Masked` wrapper's `layer.0`."""
    key = key.replace("encoder.", "").replace(".res.0", ".res")
    if ".conv" not in key:
        parts = key.split(".")
        key = ".".join(parts[:3] + ["layer", "0"] + parts[3:])
    return key


def load_quartznet_weights(encoder: nn.Module, decoder: nn.Module, weights_path: str):
    """model_weights.ckpt of a .nemo archive -> encoder / decoder (strict)."""
    sd = torch.load(weights_path, map_location="cpu")
    encoder.load_state_dict({fix_encoder_name(k): v for k, v in sd.items() if "encoder" in k}, strict=True)
    decoder.load_state_dict({k.replace("decoder.decoder_layers.0.", ""): v for k, v in sd.items() if "decoder" in k},
                            strict=True)


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
            tar.extractall(tmp, filter="data")      # untrusted archive: no absolute paths, links out of tmp, devices
        cfg = next(Path(tmp).rglob("model_config.yaml"))
        wts = next(Path(tmp).rglob("model_weights.ckpt"))
        encoder, audio_transform, text_transform = load_components_from_quartznet_config(cfg, augment_params)
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
