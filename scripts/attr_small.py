"""Which Python lines issue the small torch kernels (copies, fills) of one training step:
`attr_small.py f32|bf16` prints aten::copy_ / fill_ / zero_ calls grouped by call stack."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import net_classification, optim
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
from torch.profiler import profile, ProfilerActivity

DEV = "cuda:0"
dtype = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
torch.manual_seed(0)
net = net_classification.Net(softmax_classes=20).to(DEV).train()
opt = optim.Adam([p for p in net.parameters()], 1e-4, eps=1e-4, weight_decay=1e-6)
crit = CrossEntropyLoss2d()
x = torch.rand(6, 3, 512, 1024, device=DEV).to(dtype)
lab = torch.randint(-1, 20, (6, 512, 1024), device=DEV)


def step():
    opt.zero_grad(set_to_none=True)
    out, _ = net(x)
    crit(out, lab).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    st = [f for f in ev.stack if "erfnet_pytorch_amd" in f or "attr_small" in f][:2]
    key = (ev.name, " <- ".join(f.split("/")[-1] for f in st))
    rows[key] = rows.get(key, 0) + 1
for (name, st), n in sorted(rows.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} {name:40s} {st}")
