"""Worker of tests/test_gpu_ddp.py (started by torch.distributed.run, one process per rank; the ranks share the
visible GPUs and reduce over gloo when there are fewer GPUs than ranks).

Checks the data-parallel contract of hgvae_amd.distributed: every rank runs the reference train step on its own shard
(BatchNorm statistics over the local nodes), the flat gradient bucket is summed over ranks and divided by the world size.
That must equal a single process that runs the same shards one after the other, each from the same weights with
BatchNorm statistics per shard, and averages the gradients.  Rank 0 prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shard(rank):
    from graphs import path_batch
    gen = torch.Generator().manual_seed(100 + rank)
    ei, n, batch = path_batch([60 + 7 * rank, 41, 25 + rank])
    return torch.randn(n, 12, generator=gen), ei, batch, gen


def tape_for(n, gen):
    from test_gpu_model import skipcon_tape
    return skipcon_tape(n, 32, 4, 10, 0.5, gen)


def run_shard(model, bucket, rank, dev):
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import train_step_loss
    x, ei, batch, gen = shard(rank)
    bucket.zero()
    with use_noise(NoiseTape(tape_for(x.size(0), gen))):
        loss = train_step_loss(model, x.to(dev), ei.to(dev), batch.to(dev), 1.0)
    loss.backward()
    return loss.detach()


def main():
    from hgvae_amd.distributed import FlatGradBucket, barrier, init_from_env
    from hgvae_amd.vae import SkipconVAE
    import torch.distributed as dist
    rank, world, local_rank = init_from_env()
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(0)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    bucket = FlatGradBucket(model.parameters())
    run_shard(model, bucket, rank, dev)
    bucket.all_reduce_mean()
    torch.cuda.synchronize()
    # the benchmark's sharded workload: graph g comes from seed base + g, so the ranks' shards put together are the
    # graphs one process would have drawn (per-graph node count and feature sum, summed over the ranks)
    from hgvae_amd import synthetic
    from hgvae_amd.distributed import all_reduce_any, shard_range
    total = 2 * world + 1
    ids = shard_range(total, rank, world)
    kw = dict(nodes=(30, 50), mean_degree=4, f_in=14)
    part = synthetic.protein_like_batch_by_index(ids, 1000, dev, **kw)
    marks = torch.zeros(total, 3, dtype=torch.float64, device=dev)
    pp = part.ptr.tolist()
    for k, g in enumerate(ids):
        blk = part.x[pp[k]:pp[k + 1]].double()
        marks[g] = torch.stack([torch.tensor(float(pp[k + 1] - pp[k]), device=dev, dtype=torch.float64), blk.sum(), blk.square().sum()])
    all_reduce_any(marks)
    union_ok = True
    if rank == 0:
        whole = synthetic.protein_like_batch_by_index(range(total), 1000, dev, **kw)
        wp = whole.ptr.tolist()
        want = torch.stack([torch.stack([torch.tensor(float(wp[g + 1] - wp[g]), device=dev, dtype=torch.float64),
                                         whole.x[wp[g]:wp[g + 1]].double().sum(), whole.x[wp[g]:wp[g + 1]].double().square().sum()])
                            for g in range(total)])
        union_ok = bool(torch.equal(marks, want))
    ok, worst = True, 0.0
    if rank == 0:
        ref = SkipconVAE(12, 32, 4).to(dev).train()
        rb = FlatGradBucket(ref.parameters())
        acc = torch.zeros_like(rb.flat)
        for r in range(world):                     # same weights, fresh BatchNorm buffers for every shard
            ref.load_state_dict(state0)
            run_shard(ref, rb, r, dev)
            rb.gather()
            acc += rb.flat
            if r == 0:
                bufs_equal = all(torch.equal(a, b) for a, b in zip(model.buffers(), ref.buffers()))
        acc /= world
        worst = float((acc - bucket.flat).abs().max())
        scale = float(acc.abs().max())
        ok = worst <= 1e-6 * scale and bufs_equal and union_ok
        print(json.dumps({"world": world, "sharded_workload_union_equals_one_process_batch": union_ok, "backend": dist.get_backend(), "max_abs_diff": worst, "grad_scale": scale,
                          "rank0_bn_buffers_equal_single_process": bufs_equal, "ok": ok}), flush=True)
    barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
