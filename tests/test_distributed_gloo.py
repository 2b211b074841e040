"""world_size-2 data-parallel path on CPU (gloo): the flat gradient bucket, zero-filled unused
parameters, and equality with the single-process gradient on the union batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 3)
        self.unused = torch.nn.Linear(2, 2)  # like the SAGPooling scorer: never reaches the loss

    def forward(self, x):
        return self.a(x).pow(2).sum(1).mean()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hgvae_amd.distributed import FlatGradBucket, all_reduce_scalar_mean, init_from_env
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = Toy()
    bucket = FlatGradBucket(model.parameters())
    x = torch.randn(8, 4, generator=torch.Generator().manual_seed(1))
    shard = x[rank * 4:(rank + 1) * 4]
    grads_seen = []
    for step in range(2):  # second step checks that zero() re-arms the views
        if step == 0:
            bucket.zero()
        else:
            model.zero_grad(set_to_none=True)  # a foreign zero_grad must not break the bucket
        loss = model(shard)
        loss.backward()
        bucket.all_reduce_mean()
        grads_seen.append(bucket.flat.clone())
    assert torch.equal(grads_seen[0], grads_seen[1])
    for p, v in zip(bucket.params, bucket.views):
        assert p.grad.data_ptr() == v.data_ptr()
    mean_loss = all_reduce_scalar_mean(loss.detach())
    if rank == 0:
        torch.save({"flat": bucket.flat, "loss": mean_loss}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2(tmp_path):
    port, out = _free_port(), str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    torch.manual_seed(0)
    model = Toy()
    x = torch.randn(8, 4, generator=torch.Generator().manual_seed(1))
    # mean over the union batch = mean of the two shard means (equal shard sizes)
    loss = model(x)
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in model.parameters()])
    assert torch.allclose(got["flat"], ref, atol=1e-6)
    assert torch.allclose(got["loss"], loss.detach(), atol=1e-6)
    n_unused = sum(p.numel() for p in model.unused.parameters())
    assert float(got["flat"][-n_unused:].abs().max()) == 0.0


def test_flat_bucket_single_process_is_identity():
    from hgvae_amd.distributed import FlatGradBucket
    torch.manual_seed(0)
    model = Toy()
    bucket = FlatGradBucket(model.parameters())
    bucket.zero()
    model(torch.randn(5, 4)).backward()
    before = bucket.flat.clone()
    bucket.all_reduce_mean()
    assert torch.equal(before, bucket.flat) and bucket.world == 1


class ToyBN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 3)
        self.bn = torch.nn.BatchNorm1d(3)

    def forward(self, x):
        return self.bn(self.lin(x)).pow(2).mean()


def _ckpt_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hgvae_amd.distributed import init_from_env, save_checkpoint
    init_from_env("gloo")
    torch.manual_seed(0)
    model = ToyBN().train()
    x = torch.randn(16, 4, generator=torch.Generator().manual_seed(10 + rank)) + rank      # rank-local statistics
    model(x)
    torch.save(model.state_dict(), os.path.join(out_dir, f"own{rank}.pt"))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    save_checkpoint(model, os.path.join(out_dir, "rank0.pt"))
    save_checkpoint(model, os.path.join(out_dir, "mean.pt"), bn_buffers="mean")
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())            # saving does not modify the model
    assert os.path.exists(os.path.join(out_dir, "mean.pt"))                                 # every rank leaves after the write
    dist.destroy_process_group()


def test_rank0_checkpoint_and_batchnorm_buffer_policy(tmp_path):
    """Rank 0 writes the reference's state_dict file; BatchNorm running statistics are rank-local and the file holds
    rank 0's ('rank0') or the average over ranks ('mean'); parameters are the common ones either way."""
    mp.spawn(_ckpt_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    own = [torch.load(tmp_path / f"own{r}.pt", weights_only=True) for r in range(2)]
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    mean = torch.load(tmp_path / "mean.pt", weights_only=True)
    assert not torch.equal(own[0]["bn.running_mean"], own[1]["bn.running_mean"])
    for k in own[0]:
        assert torch.equal(r0[k], own[0][k]), k
    for k in ("lin.weight", "lin.bias", "bn.weight", "bn.bias", "bn.num_batches_tracked"):
        assert torch.equal(mean[k], own[0][k]), k
    for k in ("bn.running_mean", "bn.running_var"):
        assert torch.allclose(mean[k], (own[0][k] + own[1][k]) / 2, atol=1e-7), k
    from hgvae_amd.distributed import load_checkpoint
    fresh = ToyBN()
    load_checkpoint(fresh, str(tmp_path / "mean.pt"))
    assert torch.equal(fresh.bn.running_var, mean["bn.running_var"])


def test_sharded_workload_is_the_same_set_of_graphs_for_every_world_size():
    """The benchmark's global batch is generated by global graph index (seed = base + g): the union of the ranks'
    shards equals the one-process batch for every number of ranks, graph by graph and bit for bit."""
    from hgvae_amd import synthetic
    from hgvae_amd.distributed import shard_range
    total = 6
    whole = synthetic.protein_like_batch_by_index(range(total), 1000, "cpu", nodes=(30, 50), mean_degree=4, f_in=14)
    ptr = whole.ptr.tolist()
    for world in (1, 2, 3, 4):
        covered = []
        for rank in range(world):
            ids = shard_range(total, rank, world)
            part = synthetic.protein_like_batch_by_index(ids, 1000, "cpu", nodes=(30, 50), mean_degree=4, f_in=14)
            lo, hi = ptr[ids.start], ptr[ids.stop]
            assert torch.equal(part.x, whole.x[lo:hi])
            m = (whole.edge_index[0] >= lo) & (whole.edge_index[0] < hi)
            assert torch.equal(part.edge_index, whole.edge_index[:, m] - lo)
            covered += list(ids)
        assert covered == list(range(total))
