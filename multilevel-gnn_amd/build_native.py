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


def _object_stale(obj, cmd):
    """An object is rebuilt when it is missing, its command line changed, or a file of its dependency list (written by
    the compiler, -MD) is newer than it."""
    if not (os.path.exists(obj) and os.path.exists(obj + ".d") and os.path.exists(obj + ".cmd")):
        return True
    if open(obj + ".cmd").read() != " ".join(cmd):
        return True
    t = os.path.getmtime(obj)
    deps = open(obj + ".d").read().replace("\\\n", " ").split()[1:]
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps if not d.startswith("/opt/rocm"))


def build(force=False, verbose=True):
    """Compile every csrc/*.hip (in parallel, one hipcc per translation unit) and link ONE shared
    object next to the package."""
    if not force and not _stale():
        return LIB
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    jobs, pending, failed = [], [], []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + inc + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if force or _object_stale(obj, cmd):
            pending.append((obj, cmd))
    # at most one hipcc per host core at a time (a translation unit takes 10 s .. 3 min and up to 2 GB)
    width = max(1, min(len(pending), os.cpu_count() or 4))
    while pending or jobs:
        while pending and len(jobs) < width:
            obj, cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((obj, cmd, subprocess.Popen(cmd)))
        obj, cmd, proc = jobs.pop(0)
        if proc.wait() != 0:
            failed.append(cmd)
            if os.path.exists(obj):
                os.remove(obj)
        else:
            open(obj + ".cmd", "w").write(" ".join(cmd))
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    link = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
