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

    def _encode_on_device(self, items: List[str], device: torch.device):
        """Character vocabularies: the strings are uploaded once as code points and ts_encode_chars (csrc/metrics.hip) looks the
        ids up, adds the start / end tokens and pads -- instead of one torch.tensor + pad_sequence slice per utterance."""
        import numpy as np
        from .. import _lib
        v = self.vocab
        tab = getattr(self, "_cp_table", None)
        if tab is None or tab[0] != str(device):
            single = sorted((ord(t), i) for i, t in enumerate(v.itos) if len(t) == 1)
            cp = torch.tensor([c for c, _ in single], dtype=torch.int32, device=device)
            ids = torch.tensor([i for _, i in single], dtype=torch.int32, device=device)
            known = {chr(c) for c, _ in single}
            tab = self._cp_table = (str(device), cp, ids, known)
        _, cp, ids, known = tab
        if v.unknown_token is None:                       # numericalize drops out-of-vocabulary tokens (vocab.py:95-96)
            items = ["".join(ch for ch in s if ch in known) for s in items]
        rows = [np.frombuffer(s.encode("utf-32-le"), dtype="<u4").astype(np.int32) for s in items]
        off = np.zeros(len(rows) + 1, dtype=np.int32)
        off[1:] = np.cumsum([r.size for r in rows])
        flat = np.concatenate(rows + [np.zeros(1, np.int32)])
        extra = (v.start_token is not None) + (v.end_token is not None)
        s_max = max(int(max(r.size for r in rows)) + extra, 1)
        d = torch.from_numpy(np.concatenate([flat, off])).pin_memory().to(device, non_blocking=True)
        out = torch.empty(len(rows), s_max, dtype=torch.int64, device=device)
        lens = torch.empty(len(rows), dtype=torch.int64, device=device)
        st = _lib.lib().ts_encode_chars(d.data_ptr(), d[flat.size:].data_ptr(), len(rows), cp.data_ptr(), ids.data_ptr(), cp.numel(),
                                        v._unk_idx, v.stoi[v.start_token] if v.start_token is not None else -1,
                                        v.stoi[v.end_token] if v.end_token is not None else -1, v.pad_idx, s_max, out.data_ptr(),
                                        lens.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
        _lib.check(st, "ts_encode_chars")
        return out, lens

    def encode(self, items: List[str], return_length: bool = True, device=None) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        dev = torch.device(device) if device is not None else None
        if dev is not None and dev.type == "cuda" and self.tokenizer is char_tokenizer and len(items) > 0:
            batched, lengths = self._encode_on_device(items, dev)
            return (batched, lengths) if return_length else batched
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

    def _decode_plan(self):
        """How decode_collapsed joins tokens, decided once per vocabulary:
          ("chars", table[V] uint8/uint32, codec): every ordinary token is ONE character and no ordinary token contains the first character of
              a removable special string (blank / pad / start / end), so those strings can only come from their own token: the table maps an
              id straight to its final character ("▁", "|" -> " ", removable specials -> NUL, dropped) and the reference's replace / remove
              passes (text_processing/transform.py:107-120, vocab.py:124-130) cannot do anything else to the text;
          ("table", table[V + 2, L], codec, sep): any other vocabulary with tokens of <= 32 characters (BPE pieces): code-point table, one
              lookup for the batch, the reference's passes on every joined row;
          ("loop",): tokens containing NUL / the separator, or very long ones."""
        import numpy as np
        plan = getattr(self, "_dec_plan", None)
        if plan is not None and plan[0] == tuple(self.vocab.itos):
            return plan[1]
        voc = self.vocab
        toks = list(voc.itos)
        removable = {t for t in (voc.blank_token, voc.pad_token, voc.start_token, voc.end_token) if t}
        ordinary = [t for t in toks if t not in removable]
        width = max((len(t) for t in toks), default=1) or 1
        sep = "\x01"
        if any("\x00" in t or sep in t for t in toks) or any(len(t) == 0 for t in removable):
            out = ("loop",)
        elif (all(len(t) == 1 for t in ordinary) and not any(("▁" in r) or ("|" in r) for r in removable)
              and not any(r[0] in t for r in removable for t in ordinary)):
            final = [("\x00" if t in removable else (" " if t in ("▁", "|") else t)) for t in toks]
            narrow = all(ord(ch) < 256 for ch in final)
            out = ("chars", np.array([ord(ch) for ch in final], dtype=np.uint8 if narrow else np.uint32), "latin-1" if narrow else "utf-32-le")
        elif width <= 32:
            narrow = all(ord(ch) < 256 for t in toks for ch in t)
            arr = np.zeros((len(toks) + 2, width), dtype=np.uint8 if narrow else np.uint32)
            for i, t in enumerate(toks):
                arr[i, : len(t)] = [ord(ch) for ch in t]
            arr[len(toks) + 1, 0] = ord(sep)
            out = ("table", arr, "latin-1" if narrow else "utf-32-le", sep)
        else:
            out = ("loop",)
        self._dec_plan = (tuple(toks), out)
        return out

    def decode_collapsed(self, collapsed: torch.Tensor, counts: torch.Tensor) -> List[str]:
        """Strings from the output of the greedy-decode kernel (run-collapsed ids + per-row counts): `decode_prediction`'s token join for the
        whole batch as ONE table lookup + one bytes -> str decode (the per-id Python loop costs more than the GPU's whole forward pass at
        64 x 751 frames); same strings as `_ids_to_text` row by row (tests/test_host_r2.py compares the three plans with it)."""
        import numpy as np
        b_, t_ = collapsed.shape
        if (collapsed.is_cuda and counts.is_cuda and collapsed.dtype == counts.dtype == torch.int32 and collapsed.is_contiguous() and counts.is_contiguous()
                and collapsed.untyped_storage().data_ptr() == counts.untyped_storage().data_ptr()
                and counts.storage_offset() == collapsed.storage_offset() + b_ * t_ and counts.numel() == b_):
            # greedy_decode's packed layout: one device -> host copy (and one synchronisation) for both
            host = torch.as_strided(collapsed, (b_ * t_ + b_,), (1,), collapsed.storage_offset()).detach().to("cpu").numpy()
            rows, n = host[: b_ * t_].reshape(b_, t_), host[b_ * t_:]
        else:
            rows, n = collapsed.detach().to("cpu").numpy(), counts.detach().to("cpu").numpy()
        plan = self._decode_plan()
        v = len(self.vocab.itos)
        if plan[0] == "loop" or rows.size == 0:
            return [self._ids_to_text(rows[i, : n[i]].tolist()) for i in range(len(n))]
        b, t = rows.shape
        if plan[0] == "chars":
            # entries beyond a row's count are 0 (ts_greedy_decode fills them): the lookup validates every id (IndexError, as the reference's
            # itos[id] would raise), the slice cuts the row at its count
            text = np.take(plan[1], rows).tobytes().decode(plan[2])
            return [text[i * t: i * t + int(n[i])].replace("\x00", "") for i in range(b)]
        _, table, codec, sep = plan
        if int(rows.max(initial=0)) >= v or int(rows.min(initial=0)) < 0:       # (entries beyond a row's count are 0: ts_greedy_decode fills them)
            raise IndexError("decode_collapsed: id outside the vocabulary")
        ids = np.where(np.arange(t)[None, :] < n[:, None], rows, v)              # beyond the row's count: the empty token
        ids = np.concatenate([ids, np.full((b, 1), v + 1, dtype=ids.dtype)], axis=1)
        chars = table[ids].reshape(-1)
        text = chars[chars != 0].tobytes().decode(codec)
        voc = self.vocab
        return [voc.remove_special_tokens(part.replace("▁", " ").replace("|", " ")) for part in text.split(sep)[:b]]

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
