"""bench.py's own line, produced by fresh child processes (never an exec of this GPU-holding process): the N=1 schema
the driver parses, and -- when the box has two GPUs -- the N=2 launch exactly as the driver starts it
(``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2``), whose line must have the N=1 line's
schema with ``world_size == 2`` and a measured all-reduce.  Reduced sizes: this checks the plumbing, not the rate."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--graphs-per-gpu", "4", "--nodes", "3000", "--edges", "24000", "--members",
         "6000", "--extra-steps", "2"]
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "kernels"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_hbm_counter", "frac_compulsory",
            "avg_launch_ms", "algorithmic_bytes_per_launch", "frac_basis", "frac_algorithmic", "achieved_algorithmic"}


def _line(cmd, env=None):
    # (bench.py allocates device memory before it imports mlgnn: the guard-band allocator of a MLGNN_CANARY=1 run of
    # this suite cannot be installed in the child any more, and bench.py is not what that run is checking)
    env = {k: v for k, v in (env or os.environ).items() if k != "MLGNN_CANARY"}
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _keys(d, prefix=""):
    out = set()
    for k, v in d.items():
        out.add(prefix + k)
        if isinstance(v, dict) and k not in ("kernels", "traffic_source"):
            out |= _keys(v, prefix + k + ".")
    return out


@pytest.fixture(scope="module")
def line_n1():
    return _line([sys.executable, "bench.py", "--gpus", "1", "--cpu-baseline-graphs", "1"] + SMALL)


def test_bench_line_n1_schema(line_n1):
    out = line_n1
    assert CONTRACT <= set(out) and ROOFLINE <= set(out["roofline"])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "graphs/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]
    assert out["config"]["world_size"] == 1 and out["config"]["allreduce_ms"] is None
    assert out["roofline"]["kernel"].startswith("csr_aggregate_") and 0 < out["roofline"]["frac"] <= 1.0
    assert out["roofline"]["frac_basis"] in ("hbm_counter", "compulsory") and "arith" in out
    # the committed PMC bytes per launch belong to the default workload (64 x 10000 x 160000 x 128): at any other size the
    # compulsory bytes stand in (a counter figure of another batch size once gave frac = 12.6 here)
    assert out["roofline"]["frac_basis"] == "compulsory" and out["roofline"]["traffic"] is None
    assert 0 < out["roofline"]["frac_of_stream_copy"] <= 1.5 and out["roofline"]["stream_copy_GBps"] > 1000
    assert all(0 < k["frac"] <= 1.0 for k in out["roofline"].get("also", []))
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])
    # all three aggregators in the line, headline = softmax; and the in-line-build figure
    assert [a["aggr"] for a in out["also_aggr"]] == ["max", "mean"]
    for a in out["also_aggr"]:
        names = [k["kernel"] for k in a["kernels"]]
        assert names == ["csr_aggregate_bwd/%s/rank1" % a["aggr"], "csr_aggregate_fwd/%s/rank1" % a["aggr"]]
        assert a["ms_per_step"] > 0 and all(0 < k["frac"] <= 1.0 and k["frac_compulsory"] > 0 for k in a["kernels"])
    assert out["no_overlap_ms_per_step"] > 0


def test_bench_two_gpus_as_the_driver_launches_it(line_n1):
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: the 2-rank RCCL launch of bench.py needs two")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"] + SMALL, env)
    single_only = {"cpu_baseline", "also_aggr", "no_overlap_ms_per_step", "roofline.stream_copy_GBps",
                   "roofline.frac_of_stream_copy", "roofline.stream_copy_variants_GBps"}
    want = {k for k in _keys(line_n1) if not any(k == s or k.startswith(s + ".") for s in single_only)}
    assert _keys(out) == want
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["collective_backend"] == "nccl"
    assert out["config"]["global_batch"] == 8 and out["config"]["allreduce_ms"] > 0
    assert abs(out["value"] - 8 * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_two_ranks_rehearsal_on_one_gpu(line_n1, tmp_path):
    """BASELINE configs[3]'s launch rehearsed on a ONE-GPU box: ``python -m torch.distributed.run --nproc-per-node 2
    bench.py --gpus 2`` exactly as the driver starts it, both ranks on the single device, the collective over ``gloo``
    (MLGNN_BENCH_BACKEND=gloo: RCCL wants one device per rank).  Everything but the transport is the measured code path:
    rank sharding by graph id, the flat bucket, ONE all-reduce per step, the reached flags, max-over-ranks timing, rank
    0's single JSON line.  Then the data-parallel step itself: the same global batch of 8 graphs through two ranks of 4
    and through one process of 8 gives the same parameters after three optimizer steps (BCE loss: the DiffPool link loss
    is one Frobenius norm over the batch and does not decompose over ranks; reference step: train.py:38-69)."""
    env = dict(os.environ, MLGNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    small = [a for a in SMALL]
    small[small.index("--graphs-per-gpu") + 1] = "4"
    out = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + small, env)
    single_only = {"cpu_baseline", "also_aggr", "no_overlap_ms_per_step", "roofline.stream_copy_GBps",
                   "roofline.frac_of_stream_copy", "roofline.stream_copy_variants_GBps"}
    want = {k for k in _keys(line_n1) if not any(k == s or k.startswith(s + ".") for s in single_only)}
    assert _keys(out) == want
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["collective_backend"] == "gloo"
    assert out["config"]["global_batch"] == 8 and out["config"]["allreduce_ms"] > 0 and out["scaling"] == "weak"
    assert abs(out["value"] - 8 * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]
    # strong scaling at a fixed global batch: 2 ranks x 4 graphs == 1 process x 8 graphs, one optimizer step
    common = ["--global-batch", "8", "--steps", "1", "--warmup", "0", "--nodes", "3000", "--edges", "24000", "--members",
              "6000", "--loss", "bce", "--no-extras", "--no-cpu-baseline"]
    two, one = str(tmp_path / "two.pt"), str(tmp_path / "one.pt")
    o2 = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--dump-params", two] + common,
               env)
    o1 = _line([sys.executable, "bench.py", "--gpus", "1", "--dump-params", one] + common)
    assert o2["scaling"] == o1["scaling"] == "strong"
    assert o2["config"]["graphs_per_gpu"] == 4 and o1["config"]["graphs_per_gpu"] == 8
    assert o2["config"]["global_batch"] == o1["config"]["global_batch"] == 8
    d2, d1 = torch.load(two, weights_only=True), torch.load(one, weights_only=True)
    g2, g1, p2, p1 = d2["grads"], d1["grads"], d2["params"], d1["params"]
    assert g1.shape == g2.shape and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    # the averaged gradient of the two ranks IS the gradient of the global batch (mean of equal shard means)
    gmax = float(g1.abs().max())
    assert float((g2 - g1).abs().max()) <= 1e-5 * gmax, (float((g2 - g1).abs().max()), gmax)
    # ... and so are the stepped parameters, wherever Adam's first step lr g / (|g| + 1e-8) is well conditioned (an entry
    # whose gradient is at the level of the fp32 summation noise moves by an arbitrary fraction of lr under any
    # implementation -- torch's DDP included)
    solid = g1.abs() >= 1e-3 * gmax
    assert int(solid.sum()) > 1000
    assert float(((p2 - p1).abs() * solid).max()) <= 1e-5
    assert float((p2 - p1).abs().max()) <= 2.1e-3                       # nobody moved further than 2 lr apart
