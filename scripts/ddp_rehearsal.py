"""Two ranks on ONE GPU over gloo: walks the multi-rank set-up of bench.py stage by stage and says
where it stops.  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/ddp_rehearsal.py"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, exit=True)
rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
T0 = time.time()


def say(*a):
    print(f"[{rank} {time.time() - T0:6.2f}]", *a, flush=True)


dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from erfnet_pytorch_amd import net_classification, optim  # noqa: E402
from erfnet_pytorch_amd.ddp import GradientReducer  # noqa: E402
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d  # noqa: E402

B, H, W = int(os.environ.get("REH_B", "2")), 256, 512
torch.manual_seed(0)
STOCK = os.environ.get("REH_MODEL", "erfnet") == "torch"      # a stock torch model: no kernel of this library runs
if STOCK:
    layers = []
    for _ in range(8):
        layers += [torch.nn.Linear(1024, 1024), torch.nn.ReLU()]
    model = torch.nn.Sequential(*layers).to(dev).train()
else:
    model = net_classification.Net(softmax_classes=20).to(dev).train()
direct = os.environ.get("REH_DIRECT", "1") == "1"
red = GradientReducer(model, bucket_bytes=3_600_000, direct_write=direct)
say("reducer built, direct_write", direct)
red.broadcast_parameters(0)
torch.cuda.synchronize()
say("broadcast done")
g = torch.Generator().manual_seed(1000 + rank)
if STOCK:
    x = torch.rand(32768, 1024, generator=g).to(dev)
    lab = torch.randint(0, 1024, (32768,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    net = model
    model = lambda t: (net(t), None)       # noqa: E731
else:
    crit = CrossEntropyLoss2d()
    x = torch.rand(B, 3, H, W, generator=g).to(dev)
    lab = torch.randint(-1, 20, (B, H, W), generator=g).to(dev)
red.zero_grad()
out, _ = model(x)
torch.cuda.synchronize()
say("forward done")
loss = crit(out, lab)
loss.backward()
torch.cuda.synchronize()
say("backward done", float(loss))
red.finish()
torch.cuda.synchronize()
say("first finish done; buckets", red.bucket_sizes)
params = red.flatten_parameters()
opt = (torch.optim.Adam if STOCK else optim.Adam)(params, 1e-4, eps=1e-4, weight_decay=1e-6)
DEFER = os.environ.get("REH_DEFER", "0") == "1"
for step in range(int(os.environ.get("REH_STEPS", "10"))):
    red.zero_grad(defer=DEFER)
    out, _ = model(x)
    loss = crit(out, lab)
    loss.backward()
    if os.environ.get("REH_SYNC", "0") == "1":
        torch.cuda.synchronize()
    red.finish()
    opt.step()
    torch.cuda.synchronize()
    say("step", step, float(loss))
# replicas must hold the same parameters
chk = torch.stack([p.detach().double().sum() for p in params])
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
say("replicas equal:", all(torch.equal(both[0], b) for b in both))
dist.barrier()
dist.destroy_process_group()
