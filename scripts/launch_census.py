"""Census of the GPU launches of one headline training step (bench.py's step: reducer, flat parameters,
library Adam): `launch_census.py f32|bf16 [steps] [bn]` (bn: eval/erfnet.py's ERFNet(20), BatchNorm + Dropout2d).  Prints launches per step by kernel name, and for the
kernels torch itself launches (copies, fills, element-wise) the aten op, shapes and the Python frame
that asked for them."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet, net_classification, optim  # noqa: E402
from erfnet_pytorch_amd.ddp import GradientReducer  # noqa: E402
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d  # noqa: E402

DEV = "cuda:0"
dts = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = torch.bfloat16 if dts == "bf16" else torch.float32
torch.manual_seed(0)
bn = len(sys.argv) > 3 and sys.argv[3] == "bn"
model = (erfnet.ERFNet(20) if bn else net_classification.Net(softmax_classes=20)).to(DEV).train()
reducer = GradientReducer(model, bucket_bytes=3_600_000, direct_write=True)
crit = CrossEntropyLoss2d()
x = torch.rand(6, 3, 512, 1024, device=DEV).to(dtype)
lab = torch.randint(-1, 20, (6, 512, 1024), device=DEV)
reducer.zero_grad()
out0 = model(x)
crit(out0 if bn else out0[0], lab).backward()
reducer.finish()
opt = optim.Adam(reducer.flatten_parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)


def step():
    reducer.zero_grad()
    out = model(x)
    loss = crit(out if bn else out[0], lab)
    loss.backward()
    reducer.finish()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()

events = prof.events()
kern = collections.Counter()
ktime = collections.Counter()
for e in events:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:100]] += 1
        ktime[e.name[:100]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
total = sum(kern.values())
print(f"{total / steps:.1f} GPU launches per step, {sum(ktime.values()) / steps / 1e3:.3f} ms of GPU time per step")
for name, n in sorted(kern.items(), key=lambda kv: -ktime[kv[0]]):
    print(f"  {n / steps:6.1f} x {ktime[name] / n:8.1f} us  {name}")

# aten ops that launched torch's own kernels: group by op, shapes and the innermost frame of this repo
print("\ntorch-side ops with GPU work (per step):")
groups = collections.Counter()
for e in events:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    if e.name not in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::copy_", "aten::cat", "aten::clone",
                      "aten::_to_copy", "aten::add_", "aten::mul_", "aten::div_", "aten::sum", "aten::mean", "aten::bernoulli_",
                      "aten::empty", "aten::ones", "aten::full", "aten::ones_like", "aten::zeros_like"):
        continue
    frame = ""
    for f in (e.stack or []):
        if "erfnet_pytorch_amd" in f or "bench.py" in f or "launch_census" in f:
            frame = f.split("/")[-1]
            break
    groups[(e.name, str(e.input_shapes)[:60], frame[:90])] += 1
for (name, shapes, frame), n in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"  {n / steps:6.1f}  {name:28s} {shapes:60s} {frame}")
