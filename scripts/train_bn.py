"""Training-step throughput of eval/erfnet.py's ERFNet(20) (BatchNorm in train mode, Dropout2d) at
batch 6, 512x1024 — the BatchNorm counterpart of bench.py's workload: `train_bn.py [f32|bf16]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import erfnet
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
from erfnet_pytorch_amd.ddp import GradientReducer
dev = "cuda:0"
B = 6
torch.manual_seed(0)
m = erfnet.ERFNet(20).to(dev).train()
opt = torch.optim.Adam(m.parameters(), 1e-4, eps=1e-4, weight_decay=1e-6, fused=True)
red = GradientReducer(m, direct_write=True)
crit = CrossEntropyLoss2d()
dt_ = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
x = torch.rand(B, 3, 512, 1024, device=dev).to(dt_); lab = torch.randint(-1, 20, (B, 512, 1024), device=dev)
def step():
    red.zero_grad()
    loss = crit(m(x), lab); loss.backward(); red.finish(); opt.step()
    return loss
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"ERFNet(20) BN-train + dropout: {dt*1e3:.2f} ms/step  {B/dt:.1f} images/s  loss {float(l):.4f}  (host issue time {t_issue*1e3:.2f} ms/step)")
