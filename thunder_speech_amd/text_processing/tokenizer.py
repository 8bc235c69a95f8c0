"""Tokenizers (reference: src/thunder/text_processing/tokenizer.py)."""
from __future__ import annotations

from typing import List


def char_tokenizer(text: str) -> List[str]:
    return list(text)


def word_tokenizer(text: str) -> List[str]:
    return text.split()


class BPETokenizer:
    """sentencepiece wrapper (tokenizer.py:26-32); sentencepiece is imported lazily."""

    def __init__(self, model_path: str):
        import sentencepiece as spm
        self.tokenizer = spm.SentencePieceProcessor(model_file=model_path)

    def __call__(self, text: str) -> List[str]:
        return self.tokenizer.encode(text, out_type=str)
