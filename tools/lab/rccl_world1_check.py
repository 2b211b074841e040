import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.arange(8, dtype=torch.float32, device="cuda")
dist.all_reduce(t); dist.barrier(device_ids=[0]); torch.cuda.synchronize()
print("nccl world-1 all_reduce ok", t.tolist(), dist.get_backend())
dist.destroy_process_group()
