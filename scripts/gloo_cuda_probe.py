"""Does the gloo backend reduce tensors that live on the GPU here?  Two ranks on one device:
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/gloo_cuda_probe.py"""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
dev = torch.device("cuda", 0)
x = torch.full((1000,), float(rank + 1), device=dev)
dist.broadcast(x, 0)
print(rank, "broadcast ok", float(x[0]), flush=True)
y = torch.full((1000,), float(rank + 1), device=dev)
dist.all_reduce(y)
torch.cuda.synchronize()
print(rank, "sync all_reduce ok", float(y[0]), flush=True)
z = torch.full((900000,), float(rank + 1), device=dev)
w = dist.all_reduce(z, async_op=True)
w.wait()
torch.cuda.synchronize()
print(rank, "async all_reduce ok", float(z[0]), flush=True)
works = [dist.all_reduce(t, async_op=True) for t in (x, y, z)]
for w in works:
    w.wait()
torch.cuda.synchronize()
print(rank, "three async all_reduces ok", float(z[0]), flush=True)
dist.destroy_process_group()
