#!/usr/bin/env python3
"""Register / spill / occupancy table of the kernels of one translation unit (compiler remarks, no GPU needed):
    python tools/kernel_regs.py csrc/tallgemm.hip [substring filter ...] [-DNAME=VALUE ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multilevel-gnn_amd")


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    sys.path.insert(0, PKG)
    import build_native as bn
    cmd = [bn.HIPCC] + bn.FLAGS + bn.FILE_FLAGS.get(os.path.basename(src), []) + defs + \
        ["-I" + os.path.join(ROOT, "include"), "-I" + bn.CSRC, "-c", os.path.join(PKG, src), "-o", "/dev/null",
         "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    cur, rows = None, []
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = dict(name=subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip())
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPR Spill: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    for r in rows:
        if all(f in r["name"] for f in flt):
            print("%-90s vgpr %3d agpr %3d spill %3d scratch %4d occ %d" % (
                r["name"].replace("mlgnn::", "")[:90], r.get("vgpr", -1), r.get("agpr", -1), r.get("spill", -1),
                r.get("scratch", -1), r.get("occ", -1)))
    if "error:" in err:
        print(err[-3000:])


if __name__ == "__main__":
    main()
