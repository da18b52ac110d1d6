#!/usr/bin/env python3
"""Development tool: build multilevel-gnn_amd/mlgnn/libmlgnn_<name>.so with one translation unit recompiled under extra
-D flags (the other objects come from build/obj of the regular build), for same-box A/B runs:
    python tools/build_variant.py xt1 tallgemm.hip -DMLGNN_TG_XT=1
    MLGNN_LIB=multilevel-gnn_amd/mlgnn/libmlgnn_xt1.so python tools/bench_dense.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
import build_native as bn  # noqa: E402


def main():
    name, units = sys.argv[1], [a for a in sys.argv[2:] if a.endswith(".hip")]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    bn.build(force=False, verbose=False)
    objdir = os.path.join(ROOT, "build", "obj")
    vdir = os.path.join(ROOT, "build", "obj_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in bn.sources():
        base = os.path.basename(src)
        obj = os.path.join(objdir, base[:-4] + ".o")
        if base in units:
            obj = os.path.join(vdir, base[:-4] + ".o")
            cmd = [bn.HIPCC] + bn.FLAGS + bn.FILE_FLAGS.get(base, []) + defs + \
                ["-I" + os.path.join(ROOT, "include"), "-I" + bn.CSRC, "-c", src, "-o", obj]
            subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(bn.PKG, "mlgnn", "libmlgnn_%s.so" % name)
    subprocess.check_call([bn.HIPCC, "--offload-arch=" + bn.ARCH, "-shared", "-fPIC"] + objs + ["-o", out])
    print(out)


if __name__ == "__main__":
    main()
