"""HuggingFace wav2vec2 checkpoints -> BaseCTCModule (reference API: src/thunder/huggingface/compatibility.py:45-112).

transformers is used the way the reference uses it for LOADING -- `AutoModelForCTC`, `AutoFeatureExtractor`,
`AutoTokenizer` find and parse the checkpoint (a hub name needs the network, a local directory does not) -- and its
modules keep owning the weights, so state-dict keys are the reference's.  The arithmetic runs on the HIP path:
`HuggingFaceEncoderAdapt` (huggingface/encoder.py) for the encoder, `linear_decoder` for the CTC head,
`Wav2Vec2Preprocess` for the waveform normalisation.  Nothing here falls back to the transformers forward pass."""
from __future__ import annotations

from typing import Any, Dict, Optional
from warnings import warn

from ..blocks import linear_decoder
from ..module import BaseCTCModule
from ..text_processing.transform import BatchTextTransformer
from .encoder import HuggingFaceEncoderAdapt
from .transform import Wav2Vec2Preprocess

__all__ = ["load_huggingface_checkpoint", "module_from_huggingface", "text_transform_from_tokenizer"]


def _added_tokens(tokenizer) -> list:
    """Special tokens added after the model was trained (`additional_special_tokens` in the transformers release the
    reference pins, `extra_special_tokens` in 5.x)."""
    for attr in ("additional_special_tokens", "extra_special_tokens"):
        toks = getattr(tokenizer, attr, None)
        if toks:
            return [str(t) for t in (toks.values() if isinstance(toks, dict) else toks)]
    return []


def _special(tokenizer, attr: str) -> Optional[str]:
    """A special token of the tokenizer, unless it was bolted on after training (then the model never emits it)."""
    tok = getattr(tokenizer, attr)
    return None if tok in _added_tokens(tokenizer) else tok


def text_transform_from_tokenizer(tokenizer) -> BatchTextTransformer:
    """Vocabulary in id order with the word delimiter "|" shown as a space; CTC blank = the tokenizer's pad token
    (reference `_tok_to_transform`, compatibility.py:52-62)."""
    added = set(_added_tokens(tokenizer))
    by_id = sorted(tokenizer.get_vocab().items(), key=lambda kv: kv[1])       # the logit index of a token is its id
    tokens = [" " if t == "|" else t for t, _ in by_id if t not in added]
    pad = _special(tokenizer, "pad_token")
    return BatchTextTransformer(tokens=tokens, blank_token=pad, pad_token=pad, unknown_token=_special(tokenizer, "unk_token"))


def module_from_huggingface(model, feature_extractor, tokenizer=None) -> BaseCTCModule:
    """Assemble the module from already-loaded transformers objects (what `load_huggingface_checkpoint` does after the
    three `from_pretrained` calls; also the offline entry point for randomly initialised models)."""
    cfg = model.base_model.config
    # Wav2Vec2ForCTC puts its head on output_hidden_size when the adapter is on (modeling_wav2vec2.py)
    hidden = cfg.output_hidden_size if getattr(cfg, "add_adapter", False) else cfg.hidden_size
    text_transform, decoder = None, None
    if tokenizer is not None:
        text_transform = text_transform_from_tokenizer(tokenizer)
        decoder = linear_decoder(hidden, text_transform.num_tokens, decoder_dropout=0.0)
        if hasattr(model, "lm_head"):
            decoder[2].load_state_dict(model.lm_head.state_dict())
    mask_input = bool(feature_extractor.return_attention_mask)
    module = BaseCTCModule(encoder=HuggingFaceEncoderAdapt(model.base_model, mask_input=mask_input), decoder=decoder,
                           text_transform=text_transform, audio_transform=Wav2Vec2Preprocess(mask_input=mask_input),
                           encoder_final_dimension=hidden)
    return module.eval()


def load_huggingface_checkpoint(model_name: str, **model_kwargs: Dict[str, Any]) -> BaseCTCModule:
    """`model_name`: hub identifier ("facebook/wav2vec2-large-960h") or a local directory written by `save_pretrained`;
    `model_kwargs` go to `AutoModelForCTC.from_pretrained`.  A checkpoint without a tokenizer yields a module whose
    decoder and text_transform are None, with the reference's warning."""
    from transformers import AutoFeatureExtractor, AutoModelForCTC, AutoTokenizer
    model = AutoModelForCTC.from_pretrained(model_name, **model_kwargs)
    feature_extractor = AutoFeatureExtractor.from_pretrained(model_name)
    try:
        tokenizer = AutoTokenizer.from_pretrained(model_name)
        if not hasattr(model, "lm_head"):
            raise KeyError("lm_head")
    except (OSError, KeyError):
        warn(UserWarning("Huggingface model is missing the tokenizer! decoder and text_transform were not initialized"))
        tokenizer = None
    return module_from_huggingface(model, feature_extractor, tokenizer)
