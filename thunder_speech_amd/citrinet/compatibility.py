"""Citrinet checkpoints -- API of the reference's src/thunder/citrinet/compatibility.py (CitrinetCheckpoint :30-41,
load_components_from_citrinet_config :54-111, fix_vocab :114-130, load_citrinet_checkpoint :133-176).

The `.nemo` archives live on NGC; this environment has no network, so the loader takes a file that is already on disk (or
in `~/.thunder`) and is untested against real weights.  `build_synthetic_citrinet` builds the benchmark configuration C3
(the reference constructor with NeMo's Citrinet-1024 kernel/stride lists) with default-initialised weights.
The YAML is read with PyYAML and the archive with the standard library (the reference uses omegaconf / torchaudio).
"""
from __future__ import annotations

import tarfile
import tempfile
from enum import Enum
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from ..blocks import conv1d_decoder
from ..module import BaseCTCModule
from ..quartznet.compatibility import load_quartznet_weights
from ..quartznet.transform import FilterbankFeatures
from ..text_processing.transform import BatchTextTransformer
from ..utils import BaseCheckpoint
from .blocks import CitrinetEncoder

__all__ = ["CitrinetCheckpoint", "load_components_from_citrinet_config", "fix_vocab", "load_citrinet_checkpoint",
           "build_synthetic_citrinet", "CITRINET_1024_KERNELS", "CITRINET_1024_STRIDES"]


class CitrinetCheckpoint(BaseCheckpoint):
    """Checkpoint names of the reference (values are the NGC file stems)."""
    stt_en_citrinet_256 = "stt_en_citrinet_256"
    stt_en_citrinet_512 = "stt_en_citrinet_512"
    stt_en_citrinet_1024 = "stt_en_citrinet_1024"
    stt_es_citrinet_512 = "stt_es_citrinet_512"


# NeMo's public Citrinet body layout: three groups, the first block of each group strides by 2
CITRINET_1024_KERNELS = [11, 13, 15, 17, 19, 21, 13, 15, 17, 19, 21, 23, 25, 25, 27, 29, 31, 33, 35, 37, 39]
CITRINET_1024_STRIDES = [2, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1]


def fix_vocab(vocab_tokens: List[str]) -> List[str]:
    """NeMo word-piece spelling -> sentencepiece spelling: "##ab" continues a word, anything else starts one."""
    return [tok[2:] if tok.startswith("##") else "▁" + tok for tok in vocab_tokens]


def load_components_from_citrinet_config(config_path: Union[str, Path, Dict], sentencepiece_path: Union[str, Path],
                                         augment_params: Optional[dict] = None
                                         ) -> Tuple[nn.Module, nn.Module, BatchTextTransformer]:
    """model_config.yaml of a Citrinet .nemo -> (encoder, audio_transform, text_transform)."""
    import yaml
    params = dict(augment_params or {})
    if isinstance(config_path, dict):
        conf = config_path
    else:
        with open(config_path, "r") as f:
            conf = yaml.safe_load(f)
    body = conf["encoder"]["jasper"][1:-1]                    # first entry = stem, last = 640-channel head
    encoder = CitrinetEncoder(filters=[c["filters"] for c in body], kernel_sizes=[c["kernel"][0] for c in body],
                              strides=[c["stride"][0] for c in body], dropout=params.pop("dropout", 0.0))
    pre = conf["preprocessor"]
    sr = pre["sample_rate"]
    audio_transform = FilterbankFeatures(sample_rate=sr, n_window_size=int(pre["window_size"] * sr),
                                         n_window_stride=int(pre["window_stride"] * sr), n_fft=pre["n_fft"],
                                         nfilt=pre["features"], dither=pre["dither"], **params)
    labels = conf["labels"] if "labels" in conf else conf["decoder"]["vocabulary"]
    text_transform = BatchTextTransformer(tokens=fix_vocab(list(labels)), sentencepiece_model=str(sentencepiece_path))
    return encoder, audio_transform, text_transform


def _find_tokenizer(root: Path) -> Path:
    """`tokenizer.model` (what the reference opens, citrinet/compatibility.py:156); newer NeMo archives prefix the file name with
    a content hash, so any `*tokenizer.model` is accepted."""
    found = sorted(root.rglob("tokenizer.model")) or sorted(root.rglob("*tokenizer.model"))
    if not found:
        raise FileNotFoundError("the .nemo archive holds no sentencepiece tokenizer.model")
    return found[0]


def load_citrinet_checkpoint(checkpoint: Union[str, CitrinetCheckpoint], save_folder: Optional[str] = None,
                             augment_params: Optional[dict] = None) -> BaseCTCModule:
    """Local `.nemo` file (or a checkpoint name whose file already sits in `save_folder` / `~/.thunder`) -> module."""
    if isinstance(checkpoint, CitrinetCheckpoint):
        folder = Path(save_folder) if save_folder else Path.home() / ".thunder"
        nemo_path = folder / f"{checkpoint.value}.nemo"
    else:
        nemo_path = Path(checkpoint)
    if not nemo_path.exists():
        raise FileNotFoundError(f"{nemo_path} not found; this environment has no network access to download it")
    with tempfile.TemporaryDirectory() as tmp:
        with tarfile.open(nemo_path) as tar:
            tar.extractall(tmp, filter="data")      # untrusted archive: no absolute paths, links out of tmp, devices
        root = Path(tmp)
        encoder, audio_transform, text_transform = load_components_from_citrinet_config(
            next(root.rglob("model_config.yaml")), _find_tokenizer(root), augment_params)
        decoder = conv1d_decoder(640, num_classes=text_transform.num_tokens)
        load_quartznet_weights(encoder, decoder, str(next(root.rglob("model_weights.ckpt"))))
    return BaseCTCModule(encoder=encoder, decoder=decoder, audio_transform=audio_transform, text_transform=text_transform,
                         encoder_final_dimension=640).eval()


def build_synthetic_citrinet(filters: Optional[List[int]] = None, kernel_sizes: Optional[List[int]] = None,
                             strides: Optional[List[int]] = None, n_tokens: int = 1024,
                             encoder_state: Optional[Dict[str, torch.Tensor]] = None,
                             decoder_state: Optional[Dict[str, torch.Tensor]] = None) -> BaseCTCModule:
    """Citrinet with the reference constructor (default: the 21-block 1024-channel layout of config C3) and
    caller-provided or default-initialised weights; 80 mels, 25 ms window, `n_tokens` word pieces + blank."""
    kernel_sizes = list(kernel_sizes or CITRINET_1024_KERNELS)
    strides = list(strides or CITRINET_1024_STRIDES)
    filters = list(filters or [1024] * len(kernel_sizes))
    text_transform = BatchTextTransformer(tokens=[f"▁w{i}" for i in range(n_tokens)])
    encoder = CitrinetEncoder(filters=filters, kernel_sizes=kernel_sizes, strides=strides, feat_in=80)
    decoder = conv1d_decoder(640, text_transform.num_tokens)
    if encoder_state is not None:
        encoder.load_state_dict(encoder_state, strict=True)
    if decoder_state is not None:
        decoder.load_state_dict(decoder_state, strict=True)
    audio_transform = FilterbankFeatures(n_window_size=400, n_window_stride=160, n_fft=512, nfilt=80)
    return BaseCTCModule(encoder=encoder, decoder=decoder, audio_transform=audio_transform, text_transform=text_transform,
                         encoder_final_dimension=640).eval()
