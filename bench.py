#!/usr/bin/env python3
"""GVAE train-step benchmark (graphs/s, edge-messages/s, HBM roofline of the aggregation, MFMA rate of the projections).

  python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" is the reference's train_vae body for one mini-batch (gvae/gvae_skipcon.py:163-185):
forward, second encode, mse + kl, backward, gradient all-reduce (N > 1), Adam step.

Workloads (model in brackets):
  protein256   synthetic protein-like graphs (~5k nodes, ~60k directed edges each), GLOBAL batch 256 split evenly
               over the ranks [gvae_skipcon VAE(105,138,28)] - the default; strong scaling
  protein32    the same graphs, 32 per rank (the per-GPU shard of the 8-GPU job; weak scaling)
  large        one graph per rank: 100k nodes, 2M edges, 128-d features [VAE(128,128,28)]; weak scaling
  chig32       32 chignolin-sized path graphs (138 nodes, 60 features) per rank [VAE(60,138,28)]
  attention    the chig32 graphs [gvae_skipcon_attention VAE(60,256,28,heads=1)]
  attention32  the protein32 graphs [gvae_skipcon_attention VAE(105,256,28,heads=1)]
  pool         8 path graphs of the bundled graphs_2 sizes (454 ... 15200 nodes, 6 features) per rank
               [vae_with_pool VAE(6,138,28), SAGPooling with the GraphConv scorer in forward]
  custom_pool  the same graphs [vae_custom_pool VAE(6,138,28), GCNConv-scored pooling, loss re-decodes, beta 0.5]

With N > 1 and no torch.distributed environment the script starts N worker processes itself
(python -m torch.distributed.run) before it touches the GPU; under torchrun it is a worker.
Rank 0 prints ONE JSON line.  Inputs and CSR are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
L2_GATHER_GBS = 17800.0        # row gather served by an XCD's L2: 16.8-18.8 TB/s measured chip-wide
BF16_DENSE_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak; an fp32-accurate product costs three fp16 piece products (f16x3,
                               # the default of the NT products) or six bf16 ones (bf16x6: the weight gradients)
FP32_MFMA_TFLOPS = 157.0

WORKLOADS = {
    "protein256": ("synthetic protein-like graphs (~5k nodes, ~60k directed edges each), global batch 256 split evenly "
                   "over the GPUs, gvae_skipcon VAE(105,138,28)", "strong"),
    "protein32": ("synthetic protein-like graphs (~5k nodes, ~60k directed edges each), 32 graphs per GPU "
                  "(the per-GPU shard of batch 256 on 8 GPUs), gvae_skipcon VAE(105,138,28)", "weak"),
    "large": ("one synthetic graph per GPU: 100k nodes, 2M edges, 128-d features, gvae_skipcon VAE(128,128,28)", "weak"),
    "chig32": ("32 chignolin-sized path graphs (138 nodes, 60 features) per GPU, gvae_skipcon VAE(60,138,28)", "weak"),
    "attention": ("32 chignolin-sized path graphs per GPU, gvae_skipcon_attention VAE(60,256,28,heads=1)", "weak"),
    "attention32": ("32 synthetic protein-like graphs per GPU, gvae_skipcon_attention VAE(105,256,28,heads=1)", "weak"),
    "pool": ("8 path graphs of the bundled graphs_2 sizes (454-15200 nodes, 6 features) per GPU, vae_with_pool "
             "VAE(6,138,28) with SAGPooling(GraphConv) in forward", "weak"),
    "custom_pool": ("8 path graphs of the bundled graphs_2 sizes per GPU, vae_custom_pool VAE(6,138,28), "
                    "GCNConv-scored pooling, re-decoded loss, beta 0.5", "weak"),
}
GRAPHS2_SIZES = [454, 2627, 4697, 11213, 11455, 11940, 13239, 15200]
OTHER_WORKLOADS = ["chig32", "attention", "pool", "custom_pool", "large", "protein32"]     # timed after the headline (N = 1)
CAPTURED_TOO = {"chig32", "attention", "pool", "custom_pool"}                            # a few thousand nodes: also as a HIP graph
DTYPE_NOTE = {
    "f16x3": "f32 (storage and accumulation fp32; projections emulate fp32 on the 16-bit matrix cores: forward / grad-input "
             "products from two scaled fp16 pieces per operand and three piece products (f16x3, 22 operand bits), weight "
             "gradients from three bf16 pieces and six products (bf16x6, 24 bits); fp32_mfma_ms_per_step is the same step on "
             "v_mfma_f32_16x16x4_f32)",
    "bf16x6": "f32 (storage and accumulation fp32; projections from three bf16 pieces per operand and six piece products on "
              "the bf16 matrix cores, 24 operand bits)",
    "fp32": "f32 (projections on v_mfma_f32_16x16x4_f32)",
}


def agg_gather_bytes(n, f, e):
    """Bytes one aggregation launch consumes when every message reads its neighbour row: 4FE + output + col + rowptr."""
    return 4 * f * e + 4 * f * n + 4 * e + 4 * (n + 1)


def gather_lines_per_row(f, line=128):
    """Average number of 128-byte cache lines one gathered row touches: rows of 4 f bytes sit at a stride of 4 ceil(f/4) * 4
    bytes, so their starting offsets inside a line cycle through the multiples of gcd(stride, line)."""
    import math
    stride, data = 4 * ((f + 3) // 4 * 4), 4 * f
    period = line // math.gcd(stride, line)
    return sum(((k * stride) % line + data - 1) // line + 1 for k in range(period)) / period


def agg_compulsory_bytes(n, f, e, ld=None):
    """Every distinct byte of an aggregation launch once: the feature table read, the output written, col, rowptr."""
    ld = ld or (f + 3) // 4 * 4
    return 2 * 4 * ld * n + 4 * e + 4 * (n + 1)


SOURCE_FILES = ["bench.py", "include/hgvae_hip.h", "hgvae_amd/csrc/*.hip", "hgvae_amd/csrc/*.h", "hgvae_amd/functional.py",
                "hgvae_amd/vae.py", "hgvae_amd/nn.py", "hgvae_amd/graph.py", "hgvae_amd/synthetic.py", "hgvae_amd/noise.py",
                "hgvae_amd/optim.py", "hgvae_amd/distributed.py"]


def source_hash():
    """SHA-1 over the sources that decide what one step launches and moves: the kernels, the operators and models above
    them, the workload generator and this file.  A counter summary is only valid for the sources it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for pat in SOURCE_FILES:
        for f in sorted(glob.glob(os.path.join(ROOT, pat))):
            h.update(os.path.relpath(f, ROOT).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc(workload, graphs):
    """(counter summary of this workload written by tools/pmc_bench.py, why it was refused or None).  The summary comes
    from separate rocprofv3 --pmc passes and is kept under profiles/; one measured on other sources than the present
    ones (its ``source_hash`` differs) is refused, so a kernel change without a counter refresh cannot skew the line."""
    for name in (f"pmc_{workload}_b{graphs}.json", f"pmc_{workload}.json"):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        if doc.get("source_hash") != source_hash():
            return None, f"profiles/{name} was measured on sources {doc.get('source_hash')} (now {source_hash()}): not used"
        doc["file"] = f"profiles/{name}"
        return doc, None
    return None, "no counter summary for this workload under profiles/"


def build_workload(name, rank, world, device):
    """(batch, model kind, model dims) for this rank."""
    from hgvae_amd import synthetic
    seed = 1000 + rank
    if name in ("protein256", "protein32", "attention32"):
        # graph g of the job is drawn from seed 1000 + g whatever the number of ranks: the union of the ranks' shards of
        # protein256 is the same 256 graphs for every N (rank r holds the contiguous ids shard_range(256, r, N))
        from hgvae_amd.distributed import shard_range
        ids = shard_range(256, rank, world) if name == "protein256" else range(32 * rank, 32 * rank + 32)
        batch = synthetic.protein_like_batch_by_index(ids, base_seed=1000, device=device)
        return (batch, "attention", (105, 256, 28)) if name == "attention32" else (batch, "skipcon", (105, 138, 28))
    if name == "large":
        return synthetic.local_random_graph(100_000, 20, 128, seed=seed, device=device), "skipcon", (128, 128, 28)
    if name in ("chig32", "attention"):
        batch = synthetic.path_graph_batch([138] * 32, 60, seed=seed, device=device)
        return (batch, "skipcon", (60, 138, 28)) if name == "chig32" else (batch, "attention", (60, 256, 28))
    if name in ("pool", "custom_pool"):
        return synthetic.path_graph_batch(GRAPHS2_SIZES, 6, seed=seed, device=device), name, (6, 138, 28)
    raise SystemExit(f"unknown workload {name}")


def make_model(kind, dims):
    from hgvae_amd import vae
    if kind == "skipcon":
        return vae.SkipconVAE(*dims), vae.train_step_loss
    if kind == "attention":
        return vae.SkipconAttentionVAE(*dims, heads=1), vae.train_step_loss
    if kind == "pool":
        return vae.PoolVAE(*dims), vae.train_step_loss
    model = vae.CustomPoolVAE(*dims)
    return model, (lambda m, x, ei, b, _lam: vae.custom_pool_step_loss(m, x, ei, b))


def reference_aggregations(kind):
    """Aggregations (of E messages each; GAT ones of E + N) one reference train step performs: (sage, gat)."""
    if kind == "skipcon":
        return 14, 0        # 4 convs x 2 passes forward, convs 2-4 x 2 passes backward
    if kind == "attention":
        return 10, 4        # (3 SAGE + 1 GAT) x 2 forward, (2 SAGE + 1 GAT) x 2 backward
    # pool variants: 3 convs x 2 passes forward + convs 2-3 x 2 backward + the scorer's aggregation (forward only)
    return 11, 0


def cpu_baseline(batch, kind, dims, sample_graphs, budget_s=10.0):
    """The oracle's train step (forward, second encode, loss, backward, Adam) timed on this host's cores."""
    import torch
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
    model, loss_fn = {
        "skipcon": (OM.SkipconVAE, lambda m: OM.skipcon_loss(m, sub.x, sub.edge_index, sub.batch, 1.0)[0]),
        "attention": (OM.SkipconAttentionVAE, lambda m: OM.skipcon_loss(m, sub.x, sub.edge_index, sub.batch, 1.0)[0]),
        "pool": (OM.PoolVAE, lambda m: OM.pool_loss(m, sub.x, sub.edge_index, sub.batch, 1.0)[0]),
        "custom_pool": (OM.CustomPoolVAE, lambda m: OM.custom_pool_loss(m, sub.x, sub.edge_index, sub.batch)[0]),
    }[kind]
    model = model(*dims)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()

    def step():
        opt.zero_grad()
        loss_fn(model).backward()
        opt.step()

    step()  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 10:
            break
    sage, gat = reference_aggregations(kind)
    e, nn_ = sub.edge_index.size(1), sub.x.size(0)
    return {"value": len(items) * n / dt, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{len(items)} of the {len(ptr) - 1} graphs of rank 0's batch "
                      f"({nn_} nodes, {e} edges), {n} train steps after 1 warm-up, "
                      f"{dt:.1f} s; torch CPU restatement of the PyG operator sequence",
            "edge_messages_per_s": (sage * e + gat * (e + nn_)) * n / dt}


def timed_steps(step, steps, warmup):
    """ms per step of ``step()`` on this GPU alone (no collective): warm-up, synchronise, time, synchronise."""
    import gc
    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # the small workloads are bound by the host: a cycle collection inside the few timed steps would be most of a step
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps
    finally:
        if was_enabled:
            gc.enable()


def secondary_workload(name, device, steps=5, warmup=2, captured=False):
    """One of the other configurations, timed on this GPU after the headline run: eager train steps (and, for the
    batches of a few thousand nodes, the same step replayed as a captured HIP graph)."""
    import gc
    import torch
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    from hgvae_amd.optim import FlatAdam
    batch, kind, dims = build_workload(name, 0, 1, device)
    torch.manual_seed(0)
    hgvae_amd.manual_seed(1234)
    model, loss_fn = make_model(kind, dims)
    model = model.to(device).train()
    opt = FlatAdam(model.parameters(), lr=1e-3, capturable=captured)
    bucket = opt.bucket
    graph_of(batch.edge_index, batch.x.size(0)).check()

    def step():
        bucket.zero()
        loss = loss_fn(model, batch.x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():
            loss.backward()
        opt.step()
        return loss

    if batch.x.size(0) < 100_000:          # a few ms per step: more steps, so that one slow step is not a fifth of the figure
        steps, warmup = 4 * steps, 2 * warmup
    ms = timed_steps(step, steps, warmup)
    n_graphs = batch.num_graphs
    out = {"name": name, "workload": WORKLOADS[name][0], "graphs": n_graphs, "nodes": batch.x.size(0),
           "edges": batch.edge_index.size(1), "steps": steps, "warmup": warmup, "launch": "eager",
           "ms_per_step": round(ms, 3), "value": round(n_graphs / (ms * 1e-3), 1), "unit": "graphs/s"}
    if captured:
        from hgvae_amd.graphed import GraphedTrainStep
        g = GraphedTrainStep(model, opt, batch.x, batch.edge_index, batch.batch, 1.0, loss_fn=loss_fn, bucket=bucket, seed=1234)
        ms_g = timed_steps(g, 2 * steps, warmup)
        out["captured_ms_per_step"] = round(ms_g, 3)
        out["captured_value"] = round(n_graphs / (ms_g * 1e-3), 1)
        del g
    del model, opt, bucket, batch, step
    gc.collect()
    torch.cuda.empty_cache()
    return out


def csr_build_bytes(n, e):
    """Bytes the radix-sort CSR build of one direction moves by its own model: int64 edge list in and (key, edge id)
    pairs out (24 E), one read and one write of the 8-byte pairs per 8-bit digit of a node id, and the three int32
    outputs (12 E)."""
    passes = max(1, -(-max(n - 1, 1).bit_length() // 8))
    return (passes * 8 + 24 + 12) * e, passes


def fresh_batch_cost(batch, device, reps=3):
    """What a NEW batch costs before the step can run, measured on the headline batch: the two CSR builds from the
    batch's edge_index (the headline's timed region replays one batch, so they are outside it), and the alternative
    for a device-resident data set - gathering the batch, CSR included, from a packed shard (no sort per batch)."""
    import torch
    from hgvae_amd.graph import Graph
    from hgvae_amd.shard import GraphShard
    ei, n = batch.edge_index, batch.x.size(0)
    e = ei.size(1)

    def build():
        g = Graph(ei, n)
        g.by_src
        return g

    build()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        build()
    t.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(t) * 1e3 / reps
    per_dir, passes = csr_build_bytes(n, e)
    out = {"csr_build_us": round(us, 1), "csr_build_what": "both CSRs (by target, by source) from int64 edge_index, "
           f"{passes} radix passes of 8 bits each", "csr_build_bytes": 2 * per_dir,
           "csr_build_GBps": round(2 * per_dir / (us * 1e-6) / 1e9, 1),
           "csr_build_frac_of_hbm_peak": round(2 * per_dir / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    ptr = batch.ptr.to(device)
    gid = torch.bucketize(ei[0], ptr[1:], right=True)
    eptr = torch.zeros(ptr.numel(), dtype=torch.int64, device=device)
    eptr[1:] = torch.cumsum(torch.bincount(gid, minlength=ptr.numel() - 1), 0)
    shard = GraphShard(batch.x, ei, ptr.cpu(), eptr.cpu())
    ids = list(range(ptr.numel() - 1))
    shard.batch(ids)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        shard.batch(ids)
    t.record()
    torch.cuda.synchronize()
    us_c = s.elapsed_time(t) * 1e3 / reps
    ld = (batch.x.size(1) + 3) // 4 * 4
    moved = 8 * n * ld + 64 * e + 16 * n
    out.update({"device_collate_us": round(us_c, 1),
                "device_collate_what": "features, batch vector, edge_index and both CSRs of this batch gathered from a "
                                       "device-resident packed shard (two launches + one small index upload)",
                "device_collate_bytes": moved, "device_collate_GBps": round(moved / (us_c * 1e-6) / 1e9, 1)})
    del shard
    return out


def spawn_workers(args):
    """Start one worker per GPU with torch.distributed.run and pass its output through.  Runs before anything in
    this process has touched the GPU; this process only waits."""
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="protein256", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph")
    ap.add_argument("--cpu-sample-graphs", type=int, default=8)
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of hgvae_amd.optim.FlatAdam")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the passes after the timed region (fp32-MFMA steps, fresh-batch cost, the other workloads)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_workers(args))

    import torch
    import torch.distributed as dist
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd.distributed import FlatGradBucket, all_reduce_any, barrier, init_from_env
    from hgvae_amd.graph import graph_of

    # stdout carries the one JSON line: whatever a backend prints while it connects (gloo does, from C++) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, world, local_rank = init_from_env()
        if world > 1:
            barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    backend = dist.get_backend() if world > 1 else None

    def note(msg):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    note(f"building workload {args.workload} on cuda:{local_rank}")
    batch, kind, dims = build_workload(args.workload, rank, world, device)
    n_nodes, n_edges, n_graphs = batch.x.size(0), batch.edge_index.size(1), batch.num_graphs
    torch.manual_seed(0)                  # identical initial weights on every rank
    noise_seed = 1234 + rank              # per-rank dropout / eps streams
    hgvae_amd.manual_seed(noise_seed)
    model, loss_fn = make_model(kind, dims)
    model = model.to(device).train()
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
        loss = loss_fn(model, batch.x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():      # weight gradients are added into the bucket views by the kernels
            loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        return loss

    eager_step = step
    if args.graph:
        from hgvae_amd.graphed import GraphedTrainStep
        step = GraphedTrainStep(model, opt, batch.x, batch.edge_index, batch.batch, 1.0, loss_fn=loss_fn, bucket=bucket,
                                seed=noise_seed, warmup=max(args.warmup, 1))
    note("workload and CSR ready; warm-up")
    for _ in range(args.warmup):
        step()
    # A generation-2 pass of Python's collector walks every object torch has created (~40 ms here) and,
    # depending on allocation counts, may or may not land inside a 20-step window: collect now and park the
    # survivors in the permanent generation so that the timed steps only see young-generation passes.
    import gc
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    note("warm-up done; barrier")
    barrier()
    note("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue the steps (diagnostic only)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    note(f"timed region done: {elapsed * 1e3 / args.steps:.3f} ms/step")

    # per-launch durations of the aggregation and projection kernels: HIP events on the launching stream around
    # every launch, in a pass of its own after the timed region (the events cost ~4 us per launch)
    probe, probe_steps = Fn.LaunchProbe(), 3
    if rank == 0:
        Fn.set_probe(probe)
        for _ in range(probe_steps):
            bucket.zero()
            with Fn.direct_weight_grads():
                loss_fn(model, batch.x, batch.edge_index, batch.batch, 1.0).backward()
        torch.cuda.synchronize()
        Fn.set_probe(None)

    stats = torch.tensor([float(n_graphs), float(n_edges), float(n_nodes)], dtype=torch.float64, device=device)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        all_reduce_any(tmax, dist.ReduceOp.MAX)
        all_reduce_any(stats, dist.ReduceOp.SUM)
        elapsed = tmax[0].item()
    tot_graphs, tot_edges, tot_nodes = stats.tolist()

    if rank == 0:
        step_s = elapsed / args.steps
        pmc, pmc_refused = load_pmc(args.workload, n_graphs) if world == 1 else (None, "counter summaries are per single-GPU run")
        h = dims[1]
        recs = probe.durations_ms()
        agg = [(sh, ms) for (k, sh, ms) in recs if k == "agg"]
        aggw = [(sh, ms) for (k, sh, ms) in recs if k == "agg_w"]
        at_h = [(sh, ms) for (sh, ms) in agg if sh[1] == h and sh[0] == n_nodes] or agg

        def rate(items, model_fn):
            ms = sum(m for (_, m) in items)
            return sum(model_fn(*sh) for (sh, _) in items) / (ms * 1e-3) / 1e9 if ms > 0 else None

        roof = None
        if at_h:
            us = sum(ms for (_, ms) in at_h) / len(at_h) * 1e3
            (n_r, f_r, e_r) = at_h[0][0]
            comp = agg_compulsory_bytes(n_r, f_r, e_r)
            traffic = None
            if pmc and pmc.get("aggregation", {}).get("F") == f_r:
                traffic = pmc["aggregation"]["hbm_bytes_per_launch"]
            achieved = comp / (us * 1e-6) / 1e9
            gathered = agg_gather_bytes(n_r, f_r, e_r) / (us * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": f"k_aggregate<4,false,*> forward aggregation at F={f_r}",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_over_compulsory": round(traffic / comp, 3) if traffic is not None else None,
                    "bytes": "achieved = compulsory bytes of one launch (feature table read once, output written, col, "
                             "rowptr: every distinct byte once) / average launch time; traffic = PMC FETCH_SIZE x2 + "
                             "WRITE_SIZE per launch (fabric requests: Infinity-Cache hits included), null without a "
                             "counter summary measured on these sources",
                    "pmc_file": pmc.get("file") if pmc else None, "pmc_source_hash": pmc.get("source_hash") if pmc else None,
                    "pmc_git_commit": pmc.get("git_commit") if pmc else None, "pmc_refused": pmc_refused,
                    "avg_launch_us": round(us, 2), "launches_timed": len(at_h),
                    "compulsory_bytes_per_launch": comp,
                    "gather_bytes_per_launch": agg_gather_bytes(n_r, f_r, e_r),
                    "gathered_GBps": round(gathered, 1),
                    "cache_gather_frac": round(gathered / L2_GATHER_GBS, 4),
                    "gather_lines_GBps": round(e_r * gather_lines_per_row(f_r) * 128 / (us * 1e-6) / 1e9, 1),
                    "cache_gather_lines_frac": round(e_r * gather_lines_per_row(f_r) * 128 / (us * 1e-6) / 1e9 / L2_GATHER_GBS, 4),
                    "note": "gathered_GBps = one neighbour row per message (4FE + 4FN + 4E + 4(N+1)) / time; the re-reads "
                            "are served by L2 / Infinity Cache, so it is priced against the L2 row-gather rate "
                            f"({L2_GATHER_GBS:.0f} GB/s), not against HBM; gather_lines_GBps counts the 128-byte lines the "
                            "gathered rows touch (a row of 4F bytes at a 16-byte aligned stride straddles lines), which is "
                            "what the cache moves",
                    "all_unweighted": {"launches_per_step": len(agg) // probe_steps,
                                       "avg_launch_us": round(sum(ms for (_, ms) in agg) / len(agg) * 1e3, 2),
                                       "gathered_GBps": round(rate(agg, agg_gather_bytes), 1),
                                       "compulsory_GBps": round(rate(agg, agg_compulsory_bytes), 1),
                                       "widths": sorted({sh[1] for (sh, _) in agg})}}
            if aggw:
                roof["weighted"] = {"kernel": "k_aggregate<4,true,*> (transposed / attention-weighted)",
                                    "launches_per_step": len(aggw) // probe_steps,
                                    "avg_launch_us": round(sum(ms for (_, ms) in aggw) / len(aggw) * 1e3, 2),
                                    "gathered_GBps": round(rate(aggw, lambda n, f, e: agg_gather_bytes(n, f, e) + 4 * e), 1),
                                    "compulsory_GBps": round(rate(aggw, lambda n, f, e: agg_compulsory_bytes(n, f, e) + 4 * e), 1)}
            proj = {}
            kind_now = Fn.default_projection()
            tn_kind = "f16x3" if (kind_now == "f16x3" and Fn._tn_f16) else ("bf16x6" if kind_now in ("f16x3", "bf16x6") else "fp32")
            arith = {"nt": kind_now, "tn": tn_kind}
            pieces = {k: {"f16x3": 3, "bf16x6": 6, "fp32": 16}[v] for k, v in arith.items()}     # fp32 MFMA: 1/16 of the 16-bit rate
            for kname, label in (("nt", "C[M,N] = A[M,K] W^T (forward and grad-input products)"),
                                 ("tn", "GW[N,K] = G^T A (weight gradients, reduction included)")):
                items = [(sh, ms) for (k, sh, ms) in recs if k == kname]
                if not items:
                    continue
                ms = sum(m for (_, m) in items)
                flops = sum(2.0 * a * b * c for ((a, b, c), _) in items)
                operand = sum(4.0 * (a * b + a * c) for ((a, b, c), _) in items)     # both row-count-sized operands once
                ent = {"what": label, "launches_per_step": len(items) // probe_steps,
                       "ms_per_step": round(ms / probe_steps, 3),
                       "tflops_fp32_equiv": round(flops / (ms * 1e-3) / 1e12, 1),
                       "arithmetic": arith[kname], "piece_products": pieces[kname],
                       "frac_of_split_peak": round(flops / (ms * 1e-3) / 1e12 / (BF16_DENSE_TFLOPS / pieces[kname]), 4),
                       "operand_GBps": round(operand / (ms * 1e-3) / 1e9, 1),
                       "operand_frac_of_hbm_peak": round(operand / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                if pmc and kname in pmc.get("projections", {}):
                    ent.update(pmc["projections"][kname])
                proj[kname] = ent
            roof["projections"] = proj
            if pmc and "step" in pmc:
                sb = pmc["step"]["hbm_bytes"]
                roof["step_hbm_bytes"] = sb
                roof["step_hbm_frac"] = round(sb / step_s / 1e9 / HBM_PEAK_GBS, 4)
            else:
                roof["step_hbm_bytes"] = roof["step_hbm_frac"] = None
        extras = {}
        if world == 1 and not args.headline_only:
            # same-run passes after the headline: (1) the step with every projection on the fp32 matrix-core kernels
            note("fp32-MFMA projection steps")
            Fn.set_projection("fp32")
            try:
                extras["fp32_mfma_ms_per_step"] = round(timed_steps(eager_step, 3, 1), 3)
            finally:
                Fn.set_projection(Fn.default_projection())
            note("fresh-batch cost")
            extras["fresh_batch"] = fresh_batch_cost(batch, device)
        executed = sum(sh[2] for (k, sh, _) in recs if k in ("agg", "agg_w")) / probe_steps
        sage, gat = reference_aggregations(kind)
        ref_msgs = sage * tot_edges + gat * (tot_edges + tot_nodes)
        out = {
            "metric": "GVAE train graphs/sec", "value": round(tot_graphs / step_s, 2),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": WORKLOADS[args.workload][1],
            "vs_baseline": None, "dtype": DTYPE_NOTE[Fn.default_projection()], "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload][0], "name": args.workload,
                       "graphs_per_gpu": n_graphs, "nodes_per_gpu": n_nodes, "edges_per_gpu": n_edges,
                       "global_batch": int(tot_graphs), "parallelism": f"dp{world}",
                       "collective": None if world == 1 else f"one all-reduce of the flat fp32 gradient bucket "
                                                             f"({bucket.flat.numel()} floats) per step, backend {backend}",
                       "batchnorm": "statistics and running buffers are rank-local (nodes of the local shard); "
                                    "parameters stay identical through the averaged gradients",
                       "seeds": "weights torch.manual_seed(0) on every rank; graph g of the job from seed 1000 + g "
                                "(protein workloads: the union over the ranks is the same set for every N; other "
                                "workloads 1000 + rank); dropout / eps Philox streams 1234 + rank",
                       "optimizer": "Adam lr=1e-3 (torch fused)" if args.torch_adam else "Adam lr=1e-3 (FlatAdam, torch arithmetic)",
                       "launch": "hip graph replay" if args.graph else "eager"},
            "edge_messages_per_s": round(ref_msgs / step_s, 1),
            "edge_messages_executed_per_s": round(executed * (tot_edges / n_edges) / step_s, 1),
            "edge_messages_note": f"edge_messages_per_s counts the reference's aggregations per step ({sage} x E"
                                  + (f" + {gat} x (E + N)" if gat else "") + "); executed = edges walked by the aggregation "
                                  "launches of one step here (the first convolution is evaluated once for both encode passes)",
            "loss": round(float(loss.item()), 6),
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "roofline": roof,
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline and args.cpu_sample_graphs > 0:
            out["cpu_baseline"] = cpu_baseline(batch, kind, dims, args.cpu_sample_graphs)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.headline_only:
            # (3) the other configurations, a few steps each, so that one run of this script observes all of them
            import gc
            del model, opt, bucket, graph, batch, step, eager_step, probe
            gc.collect()
            torch.cuda.empty_cache()
            others = []
            for name in OTHER_WORKLOADS:
                if name == args.workload:
                    continue
                note(f"secondary workload {name}")
                others.append(secondary_workload(name, device, captured=name in CAPTURED_TOO))
            out["workloads"] = others
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
