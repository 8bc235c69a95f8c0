"""Compile the HIP sources of thunder_speech_amd into one in-tree shared library for gfx950.

    python -m thunder_speech_amd.build          # -> thunder_speech_amd/libthunder_speech_hip.so

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with the repo
snapshot to the GPU box.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_NAME = "libthunder_speech_hip.so"
ASAN_LIB_NAME = "libthunder_speech_hip.asan.so"      # build_asan(): host code only, AddressSanitizer + UBSan
ARCH = "gfx950"


def lib_path() -> str:
    # TS_LIB_VARIANT selects a diagnostic build (tools/variants.py: same sources, extra -D flags) for A/B timing runs;
    # unset (always, outside tools/) it is the product library
    variant = os.environ.get("TS_LIB_VARIANT", "")
    return os.path.join(PKG, LIB_NAME if not variant else LIB_NAME.replace(".so", f".{variant}.so"))


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def needs_build() -> bool:
    out = lib_path()
    if not os.path.exists(out):
        return True
    deps = sources() + glob.glob(os.path.join(PKG, "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force: bool = False, verbose: bool = True, only=None) -> str:
    out = lib_path()
    if not force and not needs_build():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    variant = os.environ.get("TS_LIB_VARIANT", "")
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    if not variant:
        # a product build leaves only its own objects behind: variant objects of past A/B runs (tools/variants.py) and objects of sources that
        # no longer exist would otherwise travel to the GPU box with every snapshot
        keep = {os.path.basename(src) + ".o" for src in sources()}
        for name in os.listdir(os.path.join(PKG, "build")):
            if name.endswith(".o") and name not in keep:
                os.remove(os.path.join(PKG, "build", name))
        for stale in glob.glob(os.path.join(PKG, LIB_NAME.replace(".so", ".*.so"))):
            if os.path.basename(stale) != ASAN_LIB_NAME:
                os.remove(stale)
    for src in sources():
        # a variant build recompiles only the sources named in `only` and links the product objects of the rest
        own = not variant or only is None or os.path.basename(src) in only
        obj = os.path.join(PKG, "build", os.path.basename(src) + (f".{variant}" if variant and own else "") + ".o")
        objs.append(obj)
        if not own and os.path.exists(obj):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
               f'-DTS_BUILD_TARGET="{ARCH}"', "-c", src, "-o", obj]
        cmd[1:1] = os.environ.get("TS_CXXFLAGS", "").split()        # diagnostic builds (tools/variants.py), e.g. TS_CXXFLAGS=-DTS_STAMP
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    # no vendor math library: every product, f32 and bf16, runs on this library's own kernels
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_asan(verbose: bool = True) -> str:
    """The library with its HOST code (the extern "C" launchers: argument checks, dispatch tables, workspace arithmetic, launch geometry) under
    AddressSanitizer + UBSan: `hipcc -fsanitize=address,undefined` instruments the host side only for gfx950 without xnack+ (GPU AddressSanitizer
    is not available on the pool), the device code objects are the ordinary ones.  tests/test_capi_host.py drives the paths that return
    before a launch on the CPU build box (the reference's sanitizer story is PyTorch's own CI; this is the library's).  Load it under
    LD_PRELOAD=<asan_runtime()> (the interpreter is not instrumented)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(PKG, ASAN_LIB_NAME)
    bdir = os.path.join(PKG, "build", "asan")
    os.makedirs(bdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in [src] + glob.glob(os.path.join(PKG, "csrc", "*.hpp")) +
                                                                 glob.glob(os.path.join(ROOT, "include", "*.h"))):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O2", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-Wno-option-ignored",
               "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
               f'-DTS_BUILD_TARGET="{ARCH}"', "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc (host-only, sanitizers) failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-Wno-option-ignored", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def asan_runtime() -> str:
    """Path of the AddressSanitizer runtime the sanitized library needs pre-loaded into an un-instrumented interpreter."""
    clang = os.path.join(os.path.dirname(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), "..", "lib", "llvm", "bin", "clang")
    return subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan())
    else:
        print(build(force="--force" in sys.argv))
