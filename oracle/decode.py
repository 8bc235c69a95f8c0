"""Oracle restatement of greedy CTC decoding and the text side it touches.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference:
  src/thunder/module.py:98-100                       predict(): argmax(1) over [B, V, T']
  src/thunder/text_processing/transform.py:94-122    decode_prediction (unique_consecutive, join,
                                                     "▁"/"|" -> " ", strip special tokens); decodes ALL
                                                     frames, out_lengths are ignored (A9)
  src/thunder/text_processing/vocab.py:18-130        Vocabulary (blank appended if absent, pad = blank)
  src/thunder/text_processing/transform.py:65-91     encode (char tokenizer, pad with pad_idx)
Pinned by the reference's own known-answer tests (tests/text/test_transforms.py:41-91), restated in
tests/test_oracle_decode.py, plus tests/golden/decode.npz.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


class Vocab:
    def __init__(self, tokens: Sequence[str], blank_token: str = "<blank>", pad_token: Optional[str] = None,
                 unknown_token: Optional[str] = None, start_token: Optional[str] = None,
                 end_token: Optional[str] = None):
        self.blank_token, self.pad_token = blank_token, pad_token or blank_token
        self.unknown_token, self.start_token, self.end_token = unknown_token, start_token, end_token
        itos = list(tokens)
        for tok in (blank_token, pad_token, unknown_token, start_token, end_token):
            if tok and tok not in itos:
                itos.append(tok)
        self.itos = itos
        self.stoi = {t: i for i, t in enumerate(itos)}
        self.blank_idx = itos.index(self.blank_token)
        self.pad_idx = itos.index(self.pad_token)


def argmax_classes(logits: np.ndarray) -> np.ndarray:
    """[B, V, T'] -> [B, T'] (lowest index wins ties, as torch.argmax does on CPU)."""
    return np.argmax(logits, axis=1)


def collapse_repeats(ids: np.ndarray) -> np.ndarray:
    """torch.unique_consecutive on one row."""
    ids = np.asarray(ids)
    if ids.size == 0:
        return ids
    keep = np.concatenate(([True], ids[1:] != ids[:-1]))
    return ids[keep]


def decode_prediction(pred_ids: np.ndarray, vocab: Vocab, remove_repeated: bool = True) -> List[str]:
    out = []
    for row in np.asarray(pred_ids):
        if remove_repeated:
            row = collapse_repeats(row)
        text = "".join(vocab.itos[int(i)] for i in row)
        text = text.replace("▁", " ").replace("|", " ")
        text = text.replace(vocab.blank_token, "").replace(vocab.pad_token, "")
        if vocab.start_token is not None:
            text = text.replace(vocab.start_token, "")
        if vocab.end_token is not None:
            text = text.replace(vocab.end_token, "")
        out.append(text)
    return out


def encode_chars(texts: Sequence[str], vocab: Vocab) -> Tuple[np.ndarray, np.ndarray]:
    """Char tokenizer + numericalize + pad (transform.py:65-91, vocab.py:66-82)."""
    rows = []
    for t in texts:
        toks = list(t)
        if vocab.start_token is not None:
            toks = [vocab.start_token] + toks
        if vocab.end_token is not None:
            toks = toks + [vocab.end_token]
        if vocab.unknown_token is None:
            toks = [x for x in toks if x in vocab.stoi]
            ids = [vocab.stoi[x] for x in toks]
        else:
            unk = vocab.stoi[vocab.unknown_token]
            ids = [vocab.stoi.get(x, unk) for x in toks]
        rows.append(ids)
    lens = np.array([len(r) for r in rows], dtype=np.int64)
    out = np.full((len(rows), int(lens.max()) if len(rows) else 0), vocab.pad_idx, dtype=np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out, lens
