#!/usr/bin/env python3
"""Headline benchmark: graphs/s of one full training step (CSR build + forward + backward +
gradient all-reduce + Adam) of the 3-level GNN on synthetic Erdos-Renyi graphs of TCGA shape
(BASELINE.json configs[1]: N=10 000 nodes, E=160 000 edges, d=128 per graph, 3 GENConv layers
+ projection pooling + 2-level DiffPool, fp32, 64 graphs per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; graphs are sharded by rank (weak scaling: 64 graphs per GPU); the only
collective is ONE RCCL all-reduce of the flat gradient bucket per step.  Rank 0 prints one JSON
line.  ``roofline`` is the live HIP-event timing of the CSR aggregation kernels over the timed
region; ``cpu_baseline`` is the CPU oracle (reference-semantics op sequence) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "multilevel-gnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graphs-per-gpu", type=int, default=64)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fix the GLOBAL batch (e.g. 512, BASELINE configs[3]) and give every GPU "
                         "global/N graphs; default 0 = weak scaling with --graphs-per-gpu each")
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=160000)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--members", type=int, default=25000)
    ap.add_argument("--aggr", default="softmax")
    ap.add_argument("--pool-batches", type=int, default=4, help="distinct pre-generated batches cycled")
    ap.add_argument("--cpu-baseline-graphs", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="build each batch's CSR in line instead of one batch ahead on a second stream")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the also_aggr (max / mean) and no-overlap legs after the timed run")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--loss", choices=["full", "bce"], default="full",
                    help="bce: without the DiffPool link / entropy terms (the link loss is one Frobenius norm over the "
                         "batch, so only the BCE step of N ranks equals the single-process step on the global batch)")
    ap.add_argument("--dump-params", help="rank 0 saves {'params', 'grads'} (flat fp32 tensors: its parameters after the "
                                          "timed run, the all-reduced gradient of the last step) here")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, model_sd):
    """Oracle forward and backward, timed separately (median of 5 after one warm-up: BASELINE.md section 3 protocol), on
    a bounded sample of the SAME workload, host cores of this box."""
    import statistics
    from mlgnn import workload as W
    from oracle import workload as OW
    # the GPU box gives one GPU's job a 16-core share; more torch threads than that only thrash
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    nb = args.cpu_baseline_graphs
    match, seg = W.membership(args.nodes, args.members)
    batch = W.collate(list(range(nb)), args.nodes, args.edges, match, seg, "cpu")
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and k != "pathway_adj")
          for k, v in model_sd.items()}
    params = [v for v in sd.values() if v.requires_grad]
    fwd, bwd = [], []
    for it in range(6):
        t0 = time.perf_counter()
        loss = OW.training_loss(sd, batch, aggr=args.aggr)
        t1 = time.perf_counter()
        torch.autograd.grad(loss, params, allow_unused=True)
        t2 = time.perf_counter()
        if it:                               # iteration 0 = warm-up (allocator, thread pool)
            fwd.append(t1 - t0)
            bwd.append(t2 - t1)
    f, b = statistics.median(fwd), statistics.median(bwd)
    return {"value": nb / (f + b), "unit": "graphs/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
            "fwd_s_median": f, "bwd_s_median": b, "timed_iterations": len(fwd),
            "sample": "%d graphs of the same synthetic workload, forward and backward timed separately, median of %d "
                      "after 1 warm-up (oracle: materialised [E,d] gather -> elementwise -> scatter, no optimizer "
                      "step)" % (nb, len(fwd))}


PROFILED_WORKLOAD = "64x10000x160000x128"     # graphs per GPU x nodes x edges x hidden of the PMC passes (bench.py defaults)


def load_traffic(workload):
    """profiles/traffic.json (HBM bytes per launch from the PMC passes of tools/profile_round.sh) -- only if it was
    measured on THESE kernel sources AND at this run's workload size (bytes per launch scale with the batch); a stale
    file is reported loudly and not used."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, {"file": "profiles/traffic.json", "stale": True, "reason": "missing"}
    try:
        blob = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        return None, {"file": "profiles/traffic.json", "stale": True, "reason": "unreadable: %s" % e}
    src = dict(blob.get("_source", {}))
    src["file"] = "profiles/traffic.json"
    import build_native
    have = build_native.sources_digest()
    if src.get("kernel_sources_sha256") != have:
        src.update(stale=True, reason="kernel sources changed since the counters were collected (re-run "
                                      "tools/profile_round.sh)", kernel_sources_now=have)
        print("bench.py: profiles/traffic.json is STALE for these kernel sources -- roofline.traffic left null",
              file=sys.stderr, flush=True)
        return None, src
    src["stale"] = False
    if src.get("workload", PROFILED_WORKLOAD) != workload:
        src.update(applicable=False, reason="counters were collected at workload %s, this run is %s: compulsory bytes stand in"
                                            % (src.get("workload", PROFILED_WORKLOAD), workload))
        return None, src
    return blob, src


def stream_copy_ceiling(dev, mib=1024, reps=10):
    """On-box streaming ceiling: device-to-device copy of a buffer far larger than the 256 MiB Infinity
    Cache with the in-tree 16-byte-per-lane copy kernel (``mlgnn_stream_copy``, csrc/sage.hip -- the shape the guide's
    float4-copy figure is quoted for; ``Tensor.copy_`` goes through the runtime's blit kernel and measured 24 % lower),
    read + written bytes per second, HIP events on the launch stream, outside the timed region."""
    from mlgnn import _lib
    src = torch.empty(mib << 20, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream

    best = {}
    for name, nt in (("plain", 0), ("non_temporal", 1)):
        def copy():
            _lib.check(_lib.lib.mlgnn_stream_copy(src.data_ptr(), dst.data_ptr(), src.numel(), nt, st), "mlgnn_stream_copy")
        for _ in range(2):
            copy()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            copy()
        b.record()
        b.synchronize()
        best[name] = 2.0 * src.numel() * reps / (a.elapsed_time(b) * 1e-3) / 1e9
    # ... and the runtime's own device-to-device copy, for reference
    for _ in range(2):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    b.synchronize()
    best["tensor_copy_"] = 2.0 * src.numel() * reps / (a.elapsed_time(b) * 1e-3) / 1e9
    return max(best.values()), best


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one process per GPU.  MLGNN_BENCH_BACKEND=gloo is a rehearsal aid only (several ranks sharing
    # the single GPU of a development box); the measured configuration is nccl (= RCCL over xGMI).
    backend = os.environ.get("MLGNN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mlgnn import ops
    from mlgnn import workload as W
    from mlgnn.dist import broadcast_parameters
    from mlgnn.optim import FlatAdam

    strong = args.global_batch > 0
    if strong and args.global_batch % world != 0:
        raise SystemExit("--global-batch must be a multiple of the number of GPUs")
    B = args.global_batch // world if strong else args.graphs_per_gpu
    match, seg = W.membership(args.nodes, args.members)
    match, seg = match.to(dev), seg.to(dev)          # ONE membership table for all batches (a per-fold constant)
    pool = []
    for k in range(args.pool_batches):
        ids = [k * world * B + rank * B + i for i in range(B)]          # graph_id = step*global + rank*B + i
        pool.append(W.collate(ids, args.nodes, args.edges, match, seg, dev))
    torch.cuda.synchronize()

    from mlgnn import CSRGraph
    main_stream = torch.cuda.current_stream()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(aggr, steps, warmup, overlap, use_timer):
        """``warmup`` untimed + exactly ``steps`` timed training steps of a freshly initialised model with aggregator
        ``aggr`` -> (seconds, kernel timer summary or None, allreduce events, final loss, bucket)."""
        torch.manual_seed(1234)
        model = W.ThreeLevelGNN(hidden=args.hidden, num_layers=3, aggr=aggr, n_members=args.members).to(dev)
        broadcast_parameters(model)
        opt = FlatAdam(model, lr=1e-3)            # Adam over the flat parameter / gradient buffers: one launch per step
        bucket = opt.bucket
        # The topology work of a step (COO -> CSR, edge attributes into CSR order) runs on a second HIP stream, one
        # batch ahead of the step that consumes it -- what an input pipeline does -- so its latency-bound kernels
        # share the GPU with the previous step's HBM-bound ones.  Every step still builds its own CSR inside the timed
        # region (K steps = K builds; the first one is waited for).  overlap=False builds it in line instead.
        side_stream = torch.cuda.Stream() if overlap else None
        ahead = {}
        ar_events = None

        reuse = os.environ.get("MLGNN_BENCH_REUSE_TOPOLOGY", "0") == "1"      # experiment only: what the CSR build costs
        built = {}

        def build_topology(i):
            batch = pool[i % len(pool)]
            if reuse and (i % len(pool)) in built:
                ahead[i] = built[i % len(pool)]
                return
            with torch.cuda.stream(side_stream):
                g = CSRGraph(batch.edge_index, batch.x.shape[0])
                # (any view of the same elements is the key RankOneEdge will look up: mlgnn.graph._same_view)
                tables = g.edge_table(batch.edge_attr[:, 0].reshape(-1, 1), 1)
                g.hub_tables("dst"), g.hub_tables("src")                            # long-row tables (csrc/hub.hip)
                ev = torch.cuda.Event()
                ev.record(side_stream)
            ahead[i] = (g, tables, ev)
            built[i % len(pool)] = ahead[i]

        def step(i, last=False):
            batch = pool[i % len(pool)]
            if side_stream is None:
                batch.csr = None                   # built inside the model's forward
            else:
                if i not in ahead:
                    build_topology(i)
                g, tables, ev = ahead.pop(i)
                main_stream.wait_event(ev)
                hub_tabs = tuple(t for d in ("dst", "src") for t in (g.hub_tables(d) or ())[:3])
                for t in (g.rowptr, g.col, g.eid, g.rowptr_t, g.col_t, g.pos_t, g.eid_t) + tuple(tables) + hub_tabs:
                    t.record_stream(main_stream)
                batch.csr = g
                if not last:
                    build_topology(i + 1)
            bucket.release()
            loss = W.training_loss(model, batch, aux=(args.loss == "full"))
            loss.backward()
            bucket.collect()
            if ar_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                bucket.all_reduce_mean()
                e1.record()
                ar_events.append((e0, e1))
            else:
                bucket.all_reduce_mean()
            opt.step()
            return loss

        for i in range(warmup):
            step(i, last=(i == warmup - 1))
        timer = ops.KernelTimer() if use_timer else None
        ops.KERNEL_TIMER = timer
        fence()
        ar_events = [] if world > 1 else None
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(warmup + i, last=(i == steps - 1))
        fence()
        elapsed = time.perf_counter() - t0
        ops.KERNEL_TIMER = None
        for batch in pool:
            batch.csr = None
        return elapsed, (timer.summary() if timer is not None else None), ar_events, float(loss.detach()), bucket, model

    elapsed, summ, ar_events, final_loss, bucket, model = run(args.aggr, args.steps, args.warmup, not args.no_overlap,
                                                               not args.no_kernel_timer)
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    if rank == 0 and args.dump_params:
        torch.save({"params": torch.cat([p.detach().reshape(-1).float() for p in bucket.params]).cpu(),
                    "grads": torch.cat([v.reshape(-1).float() for v in bucket.views]).cpu()}, args.dump_params)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        graphs = args.steps * B * world
        out = {
            "metric": "graphs/sec fwd+bwd, 3-level GNN on 10k-node d=128 synthetic; HBM GB/s %peak",
            "value": graphs / elapsed, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32",
            # what "f32" means on the matrix cores: every fp32 Linear of the step is a power-of-two scaled hi/lo fp16
            # split, 3 MFMAs per product, fp32 accumulation (error bound 3 * 2^-22 per product: csrc/tallgemm.hip:1-23);
            # aggregation, norms, losses and the optimizer are plain fp32
            "arith": "fp32 storage and accumulation; Linear products as scaled fp16 hi/lo split x3 on MFMA (bound 3*2^-22 "
                     "per product), DiffPool contractions on fp32 MFMA; everything else fp32 VALU",
            "data": "synthetic",
            "config": {"workload": "configs[1]: ER graphs N=%d E=%d x%d per GPU, d=%d, 3 GENConv(%s, res+, LayerNorm) + "
                                   "projection pooling G=%d k=2 + DiffPool 146->37->10, fp32; step = CSR build%s + fwd + "
                                   "bwd + grad all-reduce + Adam" % (args.nodes, args.edges, B, args.hidden, args.aggr,
                                                                      args.members,
                                                                      "" if args.no_overlap else " (of the next batch, on a second stream)"),
                       "graphs_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "collective_backend": backend if world > 1 else None,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "allreduce_ms": (sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)) if ar_events else None,
                       "allreduce_bytes": bucket.flat_all.numel() * 4,      # gradients + one reached flag per parameter
                       "final_loss": final_loss},
        }
        blob, tsrc = load_traffic("%dx%dx%dx%d" % (B, args.nodes, args.edges, args.hidden))
        N_all, E_all = B * args.nodes, B * args.edges

        def kernel_table(summary):
            kernels = {}
            for name, d in summary.items():
                gbs = d["bytes"] / (d["avg_ms"] * 1e-3) / 1e9
                kernels[name] = {"launches": d["launches"], "avg_ms": d["avg_ms"], "algorithmic_bytes": d["bytes"],
                                 "achieved_GBps": gbs, "frac_algorithmic": gbs / HBM_PEAK_GBS}
            return kernels

        def view(kernels, name):
            """One launch of an aggregation kernel against the 8 TB/s peak, three ways: ``frac`` is the PHYSICAL fraction
            -- HBM bytes from the PMC counters / time / peak (``frac_basis`` "hbm_counter"); without counters for these
            kernel sources the compulsory bytes stand in (perfect reuse: a lower bound on what moved, "compulsory") --
            never above 1; ``frac_algorithmic`` is SURVEY 8(d)'s contract figure (one gathered row per edge, no cache
            credit), which exceeds 1 when a graph's rows are re-used out of L2 / Infinity Cache."""
            k = kernels[name]
            secs = k["avg_ms"] * 1e-3
            # counters per aggregator ("csr_aggregate_fwd/max"), else the headline's
            traffic = None
            if blob:
                parts = name.split("/")
                traffic = blob.get("/".join(parts[:2]), blob.get(parts[0]) if parts[1] == "softmax" else None)
            # perfect reuse: every node row read once and written once, indices and edge scalars once
            backward = name.startswith("csr_aggregate_bwd")
            comp = (3 if backward else 2) * N_all * args.hidden * 4 + E_all * 8 + (N_all + 1) * 4
            phys = traffic if traffic else comp
            return {"kernel": name, "avg_launch_ms": k["avg_ms"],
                    "achieved": phys / secs / 1e9, "frac": phys / secs / 1e9 / HBM_PEAK_GBS,
                    "frac_basis": "hbm_counter" if traffic else "compulsory", "traffic": traffic,
                    "algorithmic_bytes_per_launch": k["algorithmic_bytes"], "achieved_algorithmic": k["achieved_GBps"],
                    "frac_algorithmic": k["frac_algorithmic"],
                    "frac_hbm_counter": (traffic / secs / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "compulsory_bytes": comp, "frac_compulsory": comp / secs / 1e9 / HBM_PEAK_GBS}

        if summ is not None:
            kernels = kernel_table(summ)
            # dominant = the hand-written kernel with the largest total time in the timed region
            dom = max(summ, key=lambda n: summ[n]["total_ms"])
            v = view(kernels, dom)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": v["achieved"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": v["frac"], "frac_basis": v["frac_basis"], "traffic": v["traffic"],
                               "frac_hbm_counter": v["frac_hbm_counter"], "compulsory_bytes": v["compulsory_bytes"],
                               "frac_compulsory": v["frac_compulsory"], "traffic_source": tsrc,
                               "avg_launch_ms": v["avg_launch_ms"],
                               "algorithmic_bytes_per_launch": v["algorithmic_bytes_per_launch"],
                               "achieved_algorithmic": v["achieved_algorithmic"], "frac_algorithmic": v["frac_algorithmic"],
                               "note": "achieved / frac are PHYSICAL: HBM bytes per launch from the PMC counters "
                                       "(profiles/traffic.json: 2*FETCH_SIZE + WRITE_SIZE, separate passes) / the "
                                       "HIP-event launch time / 8 TB/s; when the counters are stale for these kernel "
                                       "sources the compulsory bytes stand in (frac_basis).  frac_algorithmic is SURVEY "
                                       "8(d)'s contract figure (one gathered row per edge, no cache credit) and exceeds "
                                       "1 because a graph's 5 MB of rows are re-used out of L2 / Infinity Cache"}
            other = [n for n in summ if n.startswith("csr_aggregate_") and n != dom]
            if other:
                out["roofline"]["also"] = [view(kernels, n) for n in sorted(other)]
            if world == 1:
                ceil, variants = stream_copy_ceiling(dev)
                out["roofline"]["stream_copy_GBps"] = ceil
                out["roofline"]["stream_copy_variants_GBps"] = variants
                out["roofline"]["frac_of_stream_copy"] = v["achieved"] / ceil
            out["kernels"] = kernels
        if world == 1 and not args.no_extras:
            # "aggr in {softmax, max, mean} -- report all three, headline = softmax": the same step with
            # the other aggregators, a few steps each after the timed headline run, same process, same batches
            del model, bucket
            torch.cuda.empty_cache()
            out["also_aggr"] = []
            for aggr in ("softmax", "max", "mean"):
                if aggr == args.aggr:
                    continue
                el, sm, _, _, _, _ = run(aggr, args.extra_steps, 3, not args.no_overlap, True)
                kt = kernel_table(sm)
                out["also_aggr"].append({"aggr": aggr, "steps": args.extra_steps, "ms_per_step": el / args.extra_steps * 1e3,
                                         "value": args.extra_steps * B / el, "unit": "graphs/s",
                                         "kernels": [view(kt, n) for n in sorted(kt) if n.startswith("csr_aggregate_")]})
            # the same headline step with the topology built in line (a caller without a second stream)
            el, _, _, _, _, _ = run(args.aggr, args.extra_steps, 2, args.no_overlap, False)
            out["no_overlap_ms_per_step" if not args.no_overlap else "overlap_ms_per_step"] = el / args.extra_steps * 1e3
        if world == 1 and not args.no_cpu_baseline:
            torch.manual_seed(1234)
            out["cpu_baseline"] = cpu_baseline(args, W.ThreeLevelGNN(hidden=args.hidden, num_layers=3, aggr=args.aggr,
                                                                       n_members=args.members).state_dict())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # leave nothing for interpreter finalisation to destroy next to the runtime's own teardown
    for batch in pool:
        batch.csr = None
    pool.clear()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
