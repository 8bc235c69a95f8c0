"""Build diagnostic variants of the HIP library (same sources, extra -D flags) next to the product library, so that one
GPU session can time several of them: `TS_LIB_VARIANT=<name> python tools/bench_tcs.py` loads variant <name>.

    python tools/variants.py name1=-DFLAG1,-DFLAG2 name2=-DFLAG3 ...       # recompiles tcs_kernel.hip per variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    procs = []
    for spec in argv:
        name, _, flags = spec.partition("=")
        only = "tcs_kernel.hip"
        if ":" in name:
            name, only = name.split(":")
        env = dict(os.environ, TS_LIB_VARIANT=name, TS_CXXFLAGS=" ".join(f for f in flags.split(",") if f))
        code = ("from thunder_speech_amd import build; "
                f"print(build.build(force=True, verbose=False, only={only.split('+')!r}))")
        procs.append((name, subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env)))
    bad = [n for n, p in procs if p.wait() != 0]
    if bad:
        raise SystemExit(f"variants failed: {bad}")


if __name__ == "__main__":
    main(sys.argv[1:])
