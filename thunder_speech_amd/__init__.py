"""thunder_speech_amd -- MI355X-native (gfx950) implementation of the thunder-speech acoustic-model hot path.

Mirrors the reference's plugin surface (`thunder.registry.load_pretrained`, `thunder.module.BaseCTCModule`,
`thunder.quartznet.*`, `thunder.citrinet.*`, `thunder.ctc_loss`, `thunder.text_processing`) with the same
constructor signatures and state-dict keys; the arithmetic runs in hand-written HIP kernels behind the C ABI
of include/thunder_speech_amd.h.  There is no CPU fallback: calling a compute path without the built
extension or with CPU tensors raises.
"""
__version__ = "0.1.0"
