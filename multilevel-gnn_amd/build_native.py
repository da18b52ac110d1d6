"""Build mlgnn/libmlgnn.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Kept outside the ``mlgnn`` package on purpose: importing ``mlgnn`` dlopens the library.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "mlgnn", "libmlgnn.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "mlgnn.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every csrc/*.hip into ONE shared object next to this package."""
    if not force and not _stale():
        return LIB
    # -fno-honor-nans/-infinities: plain v_max/v_min without NaN canonicalisation; the kernels use
    # finite sentinels instead of +-inf (csrc/aggregate.hip)
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-honor-nans", "-fno-honor-infinities",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + sources() + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
