"""FinetuneCTCModule -- reference API of src/thunder/finetune.py:19-88."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch
from torch import nn

from .module import BaseCTCModule
from .registry import load_pretrained
from .text_processing.transform import BatchTextTransformer


class FinetuneCTCModule(BaseCTCModule):
    def __init__(self, checkpoint_name: str, checkpoint_kwargs: Dict[str, Any] = None,
                 decoder_class: Callable[..., nn.Module] = None, decoder_kwargs: Dict[str, Any] = None,
                 tokens: List[str] = None, text_kwargs: Dict[str, Any] = None,
                 optimizer_class=torch.optim.AdamW, optimizer_kwargs: Dict = None,
                 lr_scheduler_class=None, lr_scheduler_kwargs: Dict = None):
        checkpoint_kwargs = checkpoint_kwargs or {}
        decoder_kwargs = decoder_kwargs or {}
        text_kwargs = text_kwargs or {}
        checkpoint_data = load_pretrained(checkpoint_name, **checkpoint_kwargs)
        if decoder_class is None and tokens is not None:
            raise ValueError("New tokens were specified, but the module also needs to know the decoder class to initialize properly.")
        if decoder_class is not None and tokens is None:
            raise ValueError("A new decoder was specified, but the module also needs to know the tokens to initialize properly.")
        if decoder_class is None and checkpoint_data.decoder is None:
            raise ValueError("The checkpoint does not have a decoder: both decoder_class and tokens must be given.")
        text_transform = checkpoint_data.text_transform
        decoder = checkpoint_data.decoder
        if tokens is not None:
            text_transform = BatchTextTransformer(tokens, **text_kwargs)
            decoder = decoder_class(checkpoint_data.encoder_final_dimension, text_transform.num_tokens, **decoder_kwargs)
        super().__init__(encoder=checkpoint_data.encoder, decoder=decoder, audio_transform=checkpoint_data.audio_transform,
                         text_transform=text_transform, optimizer_class=optimizer_class, optimizer_kwargs=optimizer_kwargs,
                         lr_scheduler_class=lr_scheduler_class, lr_scheduler_kwargs=lr_scheduler_kwargs,
                         encoder_final_dimension=checkpoint_data.encoder_final_dimension)
        if hasattr(self, "save_hyperparameters"):
            try:
                self.save_hyperparameters()
            except Exception:
                pass
