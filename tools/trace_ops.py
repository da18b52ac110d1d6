#!/usr/bin/env python3
"""Development tool: which ATen operators (and which lines of ours) launch the small non-mlgnn kernels of a MultilevelGNN
training step (tools/bench_tcga.py's step under torch.profiler, with stacks)."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench_tcga  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "kirc"
    # run the bench's own setup by monkey-patching its timing loop: reuse main() up to the step function
    import types
    src = open(os.path.join(ROOT, "tools", "bench_tcga.py")).read()
    src = src.replace("    for _ in range(a.warmup):\n        step()", "    globals()['STEP'] = step\n    return")
    mod = types.ModuleType("bt")
    mod.__file__ = os.path.join(ROOT, "tools", "bench_tcga.py")
    sys.argv = ["bench_tcga.py", "--shape", shape]
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    mod.main()
    step = mod.STEP
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for ka in prof.key_averages(group_by_stack_n=8):
        if not ka.key.startswith("aten::"):
            continue
        dev_us = getattr(ka, "self_device_time_total", None)
        if dev_us is None:
            dev_us = getattr(ka, "self_cuda_time_total", 0)
        if dev_us <= 0:
            continue
        frames = [f for f in (ka.stack or []) if "/root/repo" in f]
        rows.append((dev_us, ka.count, ka.key, " <- ".join(f.split("/")[-1][:60] for f in frames[:3]) or "(no repo frame)"))
    rows.sort(reverse=True)
    for dev_us, n, name, where in rows[:70]:
        print("%7.1f us %3d  %-30s %s" % (dev_us, n, name, where))


if __name__ == "__main__":
    main()
