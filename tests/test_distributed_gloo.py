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
