#!/usr/bin/env python3
"""TFLOP/s of csrc/gemm_nt.hip at the DiffPool shapes of BASELINE configs[4] (random data), next to the library GEMM."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    from mlgnn.gemm import gemm_bf16_nt
    res = []
    only = os.environ.get("BENCH_ONLY")
    shapes = [("T = A S", 4096, 1024, 4096, 1), ("T^T", 1024, 4096, 4096, 1),
              ("S S^T", 4096, 4096, 1024, 1), ("S^T [T|S]", 1024, 2048, 4096, 2),
              ("S^T Z", 1024, 256, 4096, 16), ("4096^3", 4096, 4096, 4096, 1),
              ("dS 4 segments", 4096, 1024, 3328, 1)]
    if only:
        shapes = [x for x in shapes if x[0] in only.split(",")]
    for name, M, N, K, splits in shapes:
        pad = int(os.environ.get("BENCH_PAD", "0"))
        a = torch.randn(M, K + pad, device="cuda").bfloat16()[:, :K]
        b = torch.randn(N, K + pad, device="cuda").bfloat16()[:, :K]
        slab = torch.empty((splits, M, N), device="cuda") if splits > 1 else None
        flop = 2.0 * M * N * K
        t = timed(lambda: gemm_bf16_nt([(a, b)], splits=splits, slab=slab))
        tl = timed(lambda: a @ b.t()) if not only else 1.0
        res.append({"product": name, "variant": os.environ.get("MLGNN_GEMM_VARIANT", "0"), "pad": int(os.environ.get("BENCH_PAD", "0")), "M": M, "N": N, "K": K, "splits": splits, "us": round(t * 1e6, 1),
                    "TFLOPs": round(flop / t / 1e12, 1), "frac_of_2500": round(flop / t / 2.5e15, 3),
                    "library_us": round(tl * 1e6, 1), "library_TFLOPs": round(flop / tl / 1e12, 1)})
        print(json.dumps(res[-1]), flush=True)


if __name__ == "__main__":
    main()
