"""CPU oracle for the thunder-speech acoustic-model hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain PyTorch-CPU / numpy fp32
(fp64 where noted), the algorithm of the reference hot path
(`/root/reference/src/thunder/...`, cited per function).  It is imported only by
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`, and only as the
checker / the timed CPU baseline.  The product path (`thunder_speech_amd`) never imports it and
fails loudly when the HIP extension is missing.

Parity pinning: every function here is checked against outputs of the real reference modules
(imported from /root/reference in the build container by `tests/golden/make_golden.py`), committed
as `.npz` fixtures under `tests/golden/`, plus the reference's own known-answer tests
(lengths_to_mask truth table, same-padding closed form, encode / greedy-decode cases).
Third-party arithmetic that is absent from /root/reference (torchaudio 0.12.0
`melscale_fbanks`, torch `ctc_loss`, transformers wav2vec2) is restated from its published
algorithm; see the individual module headers for what pins each one.
"""
