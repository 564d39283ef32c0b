"""torch's own DistributedDataParallel over gloo on GPU tensors, two ranks on one device, a stock
model: does overlapped gradient reduction complete here at all?"""
import faulthandler
import os
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(60, exit=True)
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
layers = []
for _ in range(8):
    layers += [torch.nn.Linear(1024, 1024), torch.nn.ReLU()]
net = torch.nn.parallel.DistributedDataParallel(torch.nn.Sequential(*layers).to(dev), device_ids=[0], bucket_cap_mb=4)
opt = torch.optim.Adam(net.parameters(), 1e-4)
x = torch.rand(32768, 1024, device=dev)
lab = torch.randint(0, 1024, (32768,), device=dev)
T0 = time.time()
for step in range(10):
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(net(x), lab)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"[{rank} {time.time() - T0:6.2f}] step {step} {float(loss):.5f}", flush=True)
dist.barrier()
dist.destroy_process_group()
