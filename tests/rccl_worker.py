"""One rank under the RCCL backend: the flat gradient bucket of a SkipconVAE step goes through a real RCCL all-reduce
(world size 1: the sum is the input), the device-pinned barrier returns, and the process group shuts down cleanly.
What a one-GPU box can show of the multi-GPU path: the library loads, initialises and reduces the bucket in place."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from graphs import local_graph
    from hgvae_amd.distributed import FlatGradBucket
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SkipconVAE(12, 24, 4).to(dev)
    ei, n, _ = local_graph(600, 6, seed=2)
    x = torch.randn(n, 12, device=dev)
    bucket = FlatGradBucket(model.parameters())
    bucket.zero()
    train_step_loss(model, x, ei.to(dev)).backward()
    bucket.gather()
    before = bucket.flat.clone()
    dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)           # RCCL, in place on the bucket
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    same = torch.equal(before, bucket.flat)
    views_alias = all(p.grad.data_ptr() >= bucket.flat.data_ptr() and
                      p.grad.data_ptr() < bucket.flat.data_ptr() + bucket.flat.numel() * 4 for p in bucket.params)
    backend = dist.get_backend()
    dist.destroy_process_group()
    print(json.dumps({"ok": bool(same and views_alias and float(before.abs().sum()) > 0), "backend": backend,
                      "floats": bucket.flat.numel()}))


if __name__ == "__main__":
    main()
