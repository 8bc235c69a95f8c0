"""Checkpoint registry -- reference API of src/thunder/registry.py:25-66.

`load_pretrained(name)` keeps the dispatch rule ("/" in the name -> HuggingFace, otherwise a registered
checkpoint enum member; KeyError for unknown names).  The loaders themselves (NeMo .nemo / HF hub import,
SURVEY 8f rank 1) need network access and are registered lazily by `thunder_speech_amd.quartznet.compatibility`
when that module is importable; synthetic-weight builders for the benchmark configs are always registered."""
from __future__ import annotations

from typing import Callable, Dict, Type, Union

from .module import BaseCTCModule

CHECKPOINT_LOAD_FUNC_TYPE = Callable[..., BaseCTCModule]
CHECKPOINT_REGISTRY: Dict[str, CHECKPOINT_LOAD_FUNC_TYPE] = {}


def register_checkpoint_enum(checkpoints: Type, load_function: CHECKPOINT_LOAD_FUNC_TYPE):
    """Register every member of an Enum of checkpoints with the function that loads it (registry.py:28-40)."""
    from functools import partial
    for checkpoint in checkpoints:
        CHECKPOINT_REGISTRY[checkpoint.name] = partial(load_function, checkpoint)


def register_checkpoint(name: str, load_function: CHECKPOINT_LOAD_FUNC_TYPE):
    CHECKPOINT_REGISTRY[name] = load_function


def load_pretrained(checkpoint: Union[str, object], **load_kwargs) -> BaseCTCModule:
    name = checkpoint if isinstance(checkpoint, str) else checkpoint.name
    if "/" in name:
        from .huggingface.compatibility import load_huggingface_checkpoint
        return load_huggingface_checkpoint(name, **load_kwargs)
    load_fn = CHECKPOINT_REGISTRY[name]          # KeyError for unknown names, as in the reference
    return load_fn(**load_kwargs)


def _register_builtin():
    from .quartznet.compatibility import QuartznetCheckpoint, load_quartznet_checkpoint, build_synthetic_quartznet
    register_checkpoint_enum(QuartznetCheckpoint, load_quartznet_checkpoint)
    register_checkpoint("QuartzNet5x5_synthetic", lambda **kw: build_synthetic_quartznet(repeat_blocks=1, **kw))
    register_checkpoint("QuartzNet15x5_synthetic", lambda **kw: build_synthetic_quartznet(repeat_blocks=3, **kw))
    from .citrinet.compatibility import CitrinetCheckpoint, load_citrinet_checkpoint, build_synthetic_citrinet
    register_checkpoint_enum(CitrinetCheckpoint, load_citrinet_checkpoint)
    register_checkpoint("Citrinet1024_synthetic", lambda **kw: build_synthetic_citrinet(**kw))


_register_builtin()
