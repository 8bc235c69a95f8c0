"""BatchTextTransformer -- reference API of src/thunder/text_processing/transform.py:22-154.

`decode_prediction` keeps the reference semantics (unique_consecutive per row, token join, "▁"/"|" -> " ",
special-token strings removed; ALL frames are decoded, A9) but moves the id tensor to the host in one copy
instead of one element at a time; `BaseCTCModule.predict` goes further and collapses on the GPU
(`decode_collapsed`)."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor, nn
from torch.nn.utils.rnn import pad_sequence

from .tokenizer import BPETokenizer, char_tokenizer
from .vocab import Vocabulary


class BatchTextTransformer(nn.Module):
    def __init__(self, tokens: List[str], blank_token: str = "<blank>", pad_token: str = None,
                 unknown_token: str = None, start_token: str = None, end_token: str = None,
                 sentencepiece_model: Optional[str] = None,
                 custom_tokenizer_function: Callable[[str], List[str]] = None):
        super().__init__()
        self.vocab = Vocabulary(tokens, blank_token, pad_token, unknown_token, start_token, end_token)
        if custom_tokenizer_function:
            self.tokenizer = custom_tokenizer_function
        elif sentencepiece_model:
            self.tokenizer = BPETokenizer(sentencepiece_model)
        else:
            self.tokenizer = char_tokenizer

    def encode(self, items: List[str], return_length: bool = True, device=None) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        encoded = [self.vocab.numericalize(self.vocab.add_special_tokens(self.tokenizer(x))) for x in items]
        batched = pad_sequence(encoded, batch_first=True, padding_value=self.vocab.pad_idx).to(device=device)
        if return_length:
            return batched, torch.LongTensor([len(it) for it in encoded]).to(device=device)
        return batched

    def _ids_to_text(self, ids) -> str:
        out = "".join(self.vocab.decode_into_text(ids))
        out = out.replace("▁", " ").replace("|", " ")
        return self.vocab.remove_special_tokens(out)

    def decode_prediction(self, predictions: torch.Tensor, remove_repeated: bool = True) -> List[str]:
        rows = predictions.detach().to("cpu")
        out_list: List[str] = []
        for element in rows:
            if remove_repeated:
                element = torch.unique_consecutive(element)
            out_list.append(self._ids_to_text(element.tolist()))
        return out_list

    def decode_collapsed(self, collapsed: torch.Tensor, counts: torch.Tensor) -> List[str]:
        """Strings from the output of the greedy-decode kernel (run-collapsed ids + per-row counts)."""
        rows, n = collapsed.detach().to("cpu"), counts.detach().to("cpu").tolist()
        return [self._ids_to_text(rows[i, : n[i]].tolist()) for i in range(len(n))]

    @classmethod
    def from_sentencepiece(cls, output_dir: str) -> "BatchTextTransformer":
        special_tokens = ["<s>", "</s>", "<pad>", "<unk>"]
        vocab = []
        with open(f"{output_dir}/tokenizer.vocab", "r") as f:
            for line in f:
                piece = line.split("\t")[0]
                if piece not in special_tokens:
                    vocab.append(piece)
        return cls(tokens=vocab, sentencepiece_model=f"{output_dir}/tokenizer.model")

    @property
    def num_tokens(self):
        return len(self.vocab.itos)
