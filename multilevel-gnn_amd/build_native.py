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


def sources_digest():
    """sha256 over everything libmlgnn.so is built from (kernel sources, headers, this file's flags): what a set of
    profiler numbers under profiles/ is valid for."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files += [os.path.join(ROOT, "include", "mlgnn.h"), os.path.abspath(__file__)]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "mlgnn.h"))
    deps.append(os.path.abspath(__file__))                       # the flags live here
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"]
# Only the two aggregation translation units: plain v_max/v_min without NaN canonicalisation; they use finite
# sentinels instead of +-inf and carry non-finite inputs through an explicit tracker (csrc/aggregate_common.h).
# Everything else is built with default NaN semantics, so NaN guards written as comparisons stay what they say.
FILE_FLAGS = {"aggregate_fwd.hip": ["-fno-honor-nans", "-fno-honor-infinities"],
              "aggregate_bwd.hip": ["-fno-honor-nans", "-fno-honor-infinities"]}


def build(force=False, verbose=True):
    """Compile every csrc/*.hip (in parallel, one hipcc per translation unit) and link ONE shared
    object next to the package."""
    if not force and not _stale():
        return LIB
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + inc + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((obj, cmd, subprocess.Popen(cmd)))
    failed = [cmd for _, cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    link = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + [obj for obj, _, _ in jobs] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
