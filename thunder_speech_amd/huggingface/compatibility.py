"""HuggingFace wav2vec2 folder (reference: src/thunder/huggingface/compatibility.py:65-112).

Built so far for config C5: the waveform normalisation (`huggingface/transform.py`, ts_w2v_preprocess).  The wav2vec2
conv feature extractor + transformer encoder kernels are not built in this round; the entry point exists so that
`load_pretrained("org/name")` dispatches like the reference and fails loudly instead of silently running a non-HIP
path."""


def load_huggingface_checkpoint(model_name: str, **model_kwargs):
    raise NotImplementedError(
        f"load_huggingface_checkpoint({model_name!r}): the wav2vec2 HIP path (conv feature extractor, MFMA "
        "attention / FFN) is scheduled after the QuartzNet/Citrinet path (see DESIGN.md); no fallback is provided.")
