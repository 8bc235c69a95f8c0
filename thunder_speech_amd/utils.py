"""Small host-side helpers: the reference's utils.py names (audio_len, get_default_cache_folder, get_files, chain_calls, BaseCheckpoint,
download_checkpoint: utils.py:32-147) plus synthetic weights for benchmarks / smoke tests and workload accounting."""
from __future__ import annotations

import functools
import math
import os
from enum import Enum
from pathlib import Path
from typing import Callable, Dict, List, Tuple, Union

import torch
from torch import nn


def audio_len(item: Union[Path, str]) -> float:
    """Length of an audio file in seconds (reference utils.py:32-42 asks torchaudio.info; torchaudio is not in this image, so PCM WAV files
    are read with the standard library and anything else is refused)."""
    import wave
    try:
        with wave.open(str(item), "rb") as f:
            return f.getnframes() / float(f.getframerate())
    except wave.Error as e:
        raise RuntimeError(f"audio_len: {item} is not a PCM WAV file ({e}); other containers need torchaudio") from e


def get_default_cache_folder() -> Path:
    folder = Path.home() / ".thunder"
    folder.mkdir(exist_ok=True)
    return folder


def get_files(directory: Union[str, Path], extension: str) -> List[Path]:
    """All files below `directory` (links followed) whose name ends with `extension`."""
    found: List[Path] = []
    for root, _, files in os.walk(directory, followlinks=True):
        found += [Path(root) / f for f in files if f.endswith(extension)]
    return found


def chain_calls(*funcs: Callable) -> Callable:
    """g = chain_calls(f1, f2, f3): g(x) == f3(f2(f1(x)))."""
    return lambda arg: functools.reduce(lambda x, f: f(x), funcs, arg)


class BaseCheckpoint(str, Enum):
    """Base class of the pretrained-checkpoint enums; `from_string` is the argparse / hydra helper of the reference (utils.py:100-121)."""

    @classmethod
    def from_string(cls, name: str) -> "BaseCheckpoint":
        try:
            return cls[name]
        except KeyError as missing:
            raise ValueError("Name provided is not a valid checkpoint") from missing


def download_checkpoint(name: BaseCheckpoint, checkpoint_folder: str = None) -> Path:
    """Path of the checkpoint file `name` in the cache folder, fetching it first when `name.value` is a URL and the file is not there yet
    (reference utils.py:124-147).  Without network access the download fails with the URL in the message; a file that is already cached is
    returned without touching the network."""
    folder = Path(checkpoint_folder) if checkpoint_folder is not None else get_default_cache_folder()
    url = str(name.value)
    if "://" not in url:
        # this package's checkpoint enums carry bare NeMo names (no network here), the reference's carry URLs: a bare name resolves to the
        # cached `<name>.nemo`, the file name load_quartznet_checkpoint / load_citrinet_checkpoint look for
        path = folder / (url if url.endswith(".nemo") else url + ".nemo")
        if not path.exists():
            raise FileNotFoundError(f"{path} not found and {url!r} is not a URL to fetch it from")
        return path
    path = folder / url.split("/")[-1]
    if not path.exists():
        import os
        import tempfile
        import urllib.request
        folder.mkdir(parents=True, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=folder, suffix=".part")       # an interrupted download must not leave a truncated file that
        os.close(fd)                                                  # later counts as cached: fetch beside the target, then rename
        try:
            urllib.request.urlretrieve(url, tmp)
            os.replace(tmp, path)
        except OSError as e:
            raise RuntimeError(f"download_checkpoint: could not fetch {url} ({e}); place the file at {path}") from e
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return path


def variance_preserving_init_(encoder: nn.Module, decoder: nn.Module = None, seed: int = 0) -> None:
    """Random weights whose activations stay O(1) through an arbitrarily deep QuartzNet/Citrinet stack, so that
    benchmarks run on realistic magnitudes (all-zero or vanishing activations let the chip clock higher and
    would flatter the numbers).  depthwise ~ N(0, 1/K), pointwise ~ N(0, 2/Cin) (one ReLU per sub-block),
    residual and main branch each scaled by 1/sqrt(2); BatchNorm: affine ~ (1 +- 0.1, +-0.1) and perturbed
    running statistics so that BN folding is exercised.  No pretrained weights exist offline."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in encoder.named_parameters():
            if name.endswith("conv.weight"):
                cout, cin_g, k = p.shape
                if cin_g == 1 and k >= 1 and cout > 1 and ".mconv." in name and _is_depthwise(encoder, name):
                    std = math.sqrt(1.0 / k)
                else:
                    std = math.sqrt(2.0 / (cin_g * k))
                    if ".res." in name:
                        std = math.sqrt(1.0 / (cin_g * k))
                p.copy_(torch.randn(p.shape, generator=g) * std)
        for name, m in encoder.named_modules():
            if isinstance(m, nn.BatchNorm1d):
                c = m.num_features
                m.weight.copy_((1.0 + 0.1 * torch.randn(c, generator=g)) / math.sqrt(2.0))
                m.bias.copy_(0.1 * torch.randn(c, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(c, generator=g))
                m.running_var.copy_(0.8 + 0.4 * torch.rand(c, generator=g))
        if decoder is not None:
            for p in decoder.parameters():
                if p.dim() >= 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (2.0 / math.sqrt(p.shape[1])))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))


def _is_depthwise(encoder: nn.Module, param_name: str) -> bool:
    mod = encoder
    for part in param_name.split(".")[:-1]:
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    return isinstance(mod, nn.Conv1d) and mod.groups == mod.in_channels and mod.groups > 1


def tcs_algorithmic_bytes(layers, batch: int, t_in: int) -> Tuple[int, int, List[Dict]]:
    """Ideal-fusion HBM bytes and FLOPs of a list of fused TcsLayer launches: each launch reads its
    input once (bf16), reads the residual input once, writes its output once; weights are L2-resident."""
    total_b, total_f, rows = 0, 0, []
    t = t_in
    for layer in layers:
        t_out = layer.out_size(t)
        by = 2 * batch * (layer.c_in * t + layer.c_out * t_out + layer.c_res * t_out)
        macs = batch * t_out * (layer.c_in * layer.c_out + layer.c_res * layer.c_out
                                + (layer.c_in * layer.kernel if layer.depthwise else 0))
        rows.append(dict(c_in=layer.c_in, c_out=layer.c_out, k=layer.kernel, bytes=by, flops=2 * macs, t_out=t_out))
        total_b += by
        total_f += 2 * macs
        t = t_out
    return total_b, total_f, rows


class GraphedForward:
    """Replays `fn(*tensors) -> tensors` from a hipGraph, one graph per input signature (shapes, dtypes, devices).

    For the inference kernels only (no autograd, static shapes, nothing that synchronises with the host).  The inputs are copied
    into the graph's static buffers before each replay; the RETURNED TENSORS ARE THE GRAPH'S OWN OUTPUT BUFFERS and are
    overwritten by the next call with the same signature -- consume them (or clone them) before calling again."""

    def __init__(self, fn, warmup: int = 2, stream=None, alias_first: bool = False):
        # `stream`: record every capture on this side stream (default: a fresh one per signature).  The library's launch arena is keyed by
        # stream (tensors.arena), so an owner that re-captures -- BaseCTCModule after a weight update -- passes one stream and reuses it.
        # `alias_first`: the graph reads the FIRST argument where the caller's tensor lives instead of a static copy (no copy per replay); the
        # signature then includes its address, and a replay only ever happens for a tensor of that shape AT that address -- whatever was
        # allocated there since.  No reference to the caller's tensor is kept (a held tensor would pin the address the caller's allocator
        # would otherwise hand out again for the next batch).
        self.fn, self.warmup, self._graphs, self._stream, self.alias_first = fn, warmup, {}, stream, alias_first

    def signature(self, *args: torch.Tensor):
        sig = tuple((tuple(a.shape), a.dtype, str(a.device)) for a in args)
        return sig + ((args[0].data_ptr(), tuple(args[0].stride())),) if self.alias_first else sig

    def has(self, key) -> bool:
        return key in self._graphs

    def count(self) -> int:
        return len(self._graphs)

    def __call__(self, *args: torch.Tensor):
        key = self.signature(*args)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [a if (self.alias_first and i == 0) else a.detach().clone() for i, a in enumerate(args)]
            side = self._stream if self._stream is not None else torch.cuda.Stream(device=args[0].device)
            side.wait_stream(torch.cuda.current_stream(args[0].device))
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(self.warmup):                     # packs weights, sizes arena buffers, sets kernel attributes
                    self.fn(*static_in)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    out = self.fn(*static_in)
            torch.cuda.current_stream(args[0].device).wait_stream(side)
            if self.alias_first:
                static_in[0] = None                              # the address is in the key; the tensor is the caller's
            entry = self._graphs[key] = (graph, static_in, out)
        graph, static_in, out = entry
        for dst, src in zip(static_in, args):
            if dst is not None:
                dst.copy_(src)                                   # bumps dst._version: host-side caches keyed on it go stale
        graph.replay()
        return out
