"""Training-step throughput of imagenet/erfnet_imagenet.py's ERFNet(1000) (encoder + classifier,
BatchNorm in train mode) at 224x224 — the ImageNet pre-training workload: `train_imagenet.py [batch] [f32|bf16]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import erfnet_imagenet
from erfnet_pytorch_amd.ddp import GradientReducer
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt_ = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
torch.manual_seed(0)
m = erfnet_imagenet.ERFNet(1000).to(dev).train()
opt = torch.optim.SGD(m.parameters(), 0.1, momentum=0.9, weight_decay=1e-4)
red = GradientReducer(m, direct_write=True)
crit = torch.nn.CrossEntropyLoss()
x = torch.randn(B, 3, 224, 224, device=dev).to(dt_); lab = torch.randint(0, 1000, (B,), device=dev)
def step():
    red.zero_grad()
    loss = crit(m(x), lab); loss.backward(); red.finish(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"erfnet_imagenet.ERFNet(1000) batch {B}: {dt*1e3:.2f} ms/step  {B/dt:.1f} images/s  loss {float(l):.4f}")
