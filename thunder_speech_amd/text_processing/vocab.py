"""Vocabulary -- same behaviour as the reference's src/thunder/text_processing/vocab.py:18-130 (host-side
string handling; not part of the GPU hot path)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn


class Vocabulary(nn.Module):
    def __init__(self, tokens: List[str], blank_token: str = "<blank>", pad_token: Optional[str] = None,
                 unknown_token: Optional[str] = None, start_token: Optional[str] = None,
                 end_token: Optional[str] = None):
        super().__init__()
        self.unknown_token = unknown_token
        self.start_token = start_token
        self.end_token = end_token
        self.blank_token = blank_token
        self.pad_token = pad_token or blank_token
        self.itos = list(tokens)
        for extra in (blank_token, pad_token, unknown_token, start_token, end_token):
            if extra and extra not in self.itos:
                self.itos = self.itos + [extra]
        self.stoi = {token: i for i, token in enumerate(self.itos)}
        self.blank_idx = self.itos.index(self.blank_token)
        self.pad_idx = self.itos.index(self.pad_token)
        self._unk_idx = self.itos.index(self.unknown_token) if self.unknown_token is not None else -1

    def numericalize(self, tokens: List[str]) -> torch.Tensor:
        if self.unknown_token is None:
            tokens = [t for t in tokens if t in self.stoi]
        return torch.tensor([self.stoi.get(it, self._unk_idx) for it in tokens], dtype=torch.long)

    def decode_into_text(self, indices: Sequence[int]) -> List[str]:
        return [self.itos[int(it)] for it in indices]

    def add_special_tokens(self, tokens: List[str]) -> List[str]:
        if self.start_token is not None:
            tokens = [self.start_token] + tokens
        if self.end_token is not None:
            tokens = tokens + [self.end_token]
        return tokens

    def remove_special_tokens(self, text: str) -> str:
        text = text.replace(self.blank_token, "").replace(self.pad_token, "")
        if self.start_token is not None:
            text = text.replace(self.start_token, "")
        if self.end_token is not None:
            text = text.replace(self.end_token, "")
        return text
