#!/usr/bin/env python3
"""GVAE train-step benchmark (graphs/s, edge-messages/s, aggregation HBM roofline).

  python bench.py --gpus N --steps K --warmup W [--workload protein256|large|chig32]

One "step" is the reference's train_vae body for one mini-batch (gvae/gvae_skipcon.py:163-185):
forward, second encode, mse + kl, backward, gradient all-reduce (N > 1), Adam step.
Default workload: the headline configuration "synthetic protein-like graphs (~5k nodes, ~60k edges each),
batch 256 over 8 GPUs" - each rank holds its 32-graph shard (weak scaling), model
VAE(105, 138, 28) of gvae_skipcon.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  The CPU baseline (the oracle = reference-equivalent PyG operator
sequence restated in torch, timed on this host's cores) runs on rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def agg_bytes(n, f, e):
    """Algorithmic bytes of one aggregation launch: one neighbour row per message, the output, col, rowptr."""
    return 4 * f * e + 4 * f * n + 4 * e + 4 * (n + 1)


def measured_traffic(workload, f):
    """HBM bytes per aggregation launch from the committed rocprofv3 PMC summary (FETCH_SIZE x2 + WRITE_SIZE,
    calibrated on a copy of known size); None when no summary exists for this workload / width."""
    path = os.path.join(ROOT, "profiles", f"agg_pmc_{workload}.json")
    try:
        d = json.load(open(path))
    except OSError:
        return None
    return d["agg_fwd"]["hbm_bytes_per_launch"] if d.get("F") == f else None


def build_workload(name, rank, device):
    from hgvae_amd import synthetic
    if name == "protein256":
        batch = synthetic.protein_like_batch(32, seed=1000 + rank, device=device)
        dims = (105, 138, 28)
    elif name == "large":
        batch = synthetic.local_random_graph(100_000, 20, 128, seed=1000 + rank, device=device)
        dims = (128, 128, 28)
    elif name == "chig32":
        batch = synthetic.path_graph_batch([138] * 32, 60, seed=1000 + rank, device=device)
        dims = (60, 138, 28)
    else:
        raise SystemExit(f"unknown workload {name}")
    return batch, dims


def cpu_baseline(batch, dims, sample_graphs, lambda_kl=1.0):
    """Time the oracle's train step (forward, second encode, loss, backward, Adam) on host cores."""
    from hgvae_amd.data import Batch, Data
    from oracle import models as OM
    ptr = batch.ptr.cpu().tolist()
    x, ei = batch.x.cpu(), batch.edge_index.cpu()
    src_graph = torch.bucketize(ei[0], torch.tensor(ptr[1:]), right=True)
    items = []
    for g in range(min(sample_graphs, len(ptr) - 1)):
        m = src_graph == g
        items.append(Data(x=x[ptr[g]:ptr[g + 1]], edge_index=ei[:, m] - ptr[g]))
    sub = Batch.from_data_list(items)
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    model = OM.SkipconVAE(*dims)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()

    def step():
        opt.zero_grad()
        loss = OM.skipcon_loss(model, sub.x, sub.edge_index, sub.batch, lambda_kl)[0]
        loss.backward()
        opt.step()

    step()  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 10:
            break
    return {"value": len(items) * n / dt, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{len(items)} of the {len(ptr) - 1} graphs of rank 0's batch "
                      f"({sub.x.size(0)} nodes, {sub.edge_index.size(1)} edges), {n} train steps after 1 warm-up, "
                      f"{dt:.1f} s; torch CPU restatement of the PyG operator sequence",
            "edge_messages_per_s": 14 * sub.edge_index.size(1) * n / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="protein256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured HIP graph (per-launch event timing is then taken "
                         "from a short eager pass after the timed region)")
    ap.add_argument("--cpu-sample-graphs", type=int, default=8)
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of hgvae_amd.optim.FlatAdam")
    args = ap.parse_args()

    import torch.distributed as dist
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd.distributed import FlatGradBucket, barrier, init_from_env
    from hgvae_amd.graph import graph_of
    from hgvae_amd.vae import SkipconVAE, train_step_loss

    rank, world, local_rank = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    def note(msg):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    note(f"building workload {args.workload} on cuda:{local_rank}")
    batch, dims = build_workload(args.workload, rank, device)
    n_nodes, n_edges, n_graphs = batch.x.size(0), batch.edge_index.size(1), batch.num_graphs
    torch.manual_seed(0)
    hgvae_amd.manual_seed(1234 + rank)
    model = SkipconVAE(*dims).to(device).train()
    if args.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=args.graph)
        bucket = FlatGradBucket(model.parameters())
    else:   # same arithmetic, one launch over the flat parameter / gradient buffers
        from hgvae_amd.optim import FlatAdam
        opt = FlatAdam(model.parameters(), lr=1e-3, capturable=args.graph)
        bucket = opt.bucket
    graph = graph_of(batch.edge_index, n_nodes)  # CSR of the batch: built once, reused by every step
    graph.check()

    def step():
        bucket.zero()
        loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():      # weight gradients are added into the bucket views by the kernels
            loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        return loss

    note("workload and CSR ready; warm-up")
    probe = Fn.LaunchProbe()
    if args.graph:
        from hgvae_amd.graphed import GraphedTrainStep
        eager_step = step
        step = GraphedTrainStep(model, opt, batch.x, batch.edge_index, batch.batch, 1.0, bucket=bucket,
                                seed=1234 + rank, warmup=max(args.warmup, 1))
    for _ in range(args.warmup):
        step()
    # A generation-2 pass of Python's collector walks every object torch has created (~40 ms here) and,
    # depending on allocation counts, may or may not land inside a 20-step window: collect now and park the
    # survivors in the permanent generation so that the timed steps only see young-generation passes.
    import gc
    gc.collect()
    gc.freeze()
    note("timed region")
    if not args.graph:
        Fn.set_probe(probe)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue the steps (diagnostic only)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.graph:   # kernels inside a replayed graph cannot be bracketed by events: time them eagerly now
        Fn.set_probe(probe)
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
    Fn.set_probe(None)

    stats = torch.tensor([elapsed, float(n_graphs), float(n_edges), float(n_nodes)], dtype=torch.float64, device=device)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = mx[0].item()
    tot_graphs, tot_edges, tot_nodes = stats[1].item(), stats[2].item(), stats[3].item()

    if rank == 0:
        # roofline of the dominant kernel k_aggregate<4,false,*> (unweighted aggregation, every width it ran at):
        # sum of algorithmic bytes / sum of launch durations, so that avg_launch_us is directly comparable with
        # the rocprofv3 --kernel-trace --stats average for that kernel name
        recs = probe.durations_ms()
        h = dims[1]
        dom = [(n, f, e, ms) for (n, f, e, w, ms) in recs if not w]
        wtd = [(n, f, e, ms) for (n, f, e, w, ms) in recs if w]
        roof = None
        if dom:
            tot_ms = sum(ms for (*_, ms) in dom)
            tot_b = sum(agg_bytes(n, f, e) for (n, f, e, ms) in dom)
            achieved = tot_b / (tot_ms * 1e-3) / 1e9
            at_h = [ms for (n, f, e, ms) in dom if f == h]
            roof = {"bound": "hbm", "kernel": "k_aggregate<4,false,*>", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": measured_traffic(args.workload, h),
                    "avg_launch_us": round(tot_ms / len(dom) * 1e3, 2), "launches": len(dom),
                    "algorithmic_bytes_per_launch": round(tot_b / len(dom)),
                    "widths": sorted({f for (n, f, e, ms) in dom}),
                    "at_hidden_width": {"F": h, "avg_launch_us": round(sum(at_h) / len(at_h) * 1e3, 2) if at_h else None,
                                        "algorithmic_bytes": agg_bytes(n_nodes, h, n_edges),
                                        "traffic_note": "traffic is the PMC-measured HBM bytes of one launch at this width"},
                    "weighted_kernel": {"kernel": "k_aggregate<4,true,*>", "launches": len(wtd),
                                        "avg_launch_us": round(sum(ms for (*_, ms) in wtd) / len(wtd) * 1e3, 2) if wtd else None,
                                        "achieved": round(sum(agg_bytes(n, f, e) + 4 * e for (n, f, e, ms) in wtd)
                                                          / (sum(ms for (*_, ms) in wtd) * 1e-3) / 1e9, 1) if wtd else None}}
        out = {
            "metric": "GVAE train graphs/sec", "value": round(tot_graphs * args.steps / elapsed, 2),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"protein256": "synthetic protein-like graphs (~5k nodes, ~60k directed edges each), "
                                                  "32 graphs per GPU (global batch 256 at 8 GPUs), gvae_skipcon VAE(105,138,28)",
                                    "large": "one synthetic graph per GPU: 100k nodes, 2M edges, 128-d features, VAE(128,128,28)",
                                    "chig32": "32 chignolin-sized path graphs (138 nodes) per GPU, VAE(60,138,28)"}[args.workload],
                       "graphs_per_gpu": n_graphs, "nodes_per_gpu": n_nodes, "edges_per_gpu": n_edges,
                       "parallelism": f"dp{world}",
                       "optimizer": "Adam lr=1e-3 (torch fused)" if args.torch_adam else "Adam lr=1e-3 (FlatAdam, torch arithmetic)",
                       "launch": "hip graph replay" if args.graph else "eager",
                       "note": "edge_messages_per_s counts the reference's 14 aggregations x E per step; the first "
                               "convolution (no stochastic layer precedes it) is evaluated once for both encode passes"},
            "edge_messages_per_s": round(14 * tot_edges * args.steps / elapsed, 1),
            "loss": round(float(loss.item()), 6),
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            if args.cpu_sample_graphs > 0:
                out["cpu_baseline"] = cpu_baseline(batch, dims, args.cpu_sample_graphs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
