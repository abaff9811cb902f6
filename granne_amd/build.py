"""Builds libgranne_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m granne_amd.build [--force]

The library is built IN-TREE (granne_amd/lib/) so that it travels with the source snapshot to
the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgranne_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    # bit-exact f32: no implicit contraction (every fused op in the kernels is an explicit fmaf),
    # IEEE-correct sqrt and divide
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return d


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_library(force=False, use_dpp=None, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    if use_dpp is None:
        use_dpp = int(os.environ.get("GRANNE_HIP_USE_DPP", "1"))
    extra = os.environ.get("GRANNE_HIP_EXTRA_FLAGS", "").split()  # kernel experiments, e.g. -DGRANNE_HIP_PQ_MERGE_MIN=99
    cmd = [HIPCC] + FLAGS + ["-DGRANNE_HIP_USE_DPP=%d" % int(use_dpp)] + extra + ["-I", INCLUDE] + _sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
