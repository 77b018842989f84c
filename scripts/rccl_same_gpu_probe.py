"""Does RCCL on this box accept two ranks on ONE device?  (torchrun --nproc-per-node 2, both on cuda:0)"""
import os
import torch
import torch.distributed as dist

dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
r = dist.get_rank()
t = torch.full((1 << 20,), float(r + 1), device="cuda:0")
out = [torch.empty_like(t) for _ in range(2)]
dist.all_gather(out, t)
torch.cuda.synchronize()
print(f"rank {r}: gathered {out[0][0].item()} {out[1][0].item()}")
dist.destroy_process_group()
