"""End-to-end timings of the bf16 path: full-resolution inference (eval_forwardTime.py loop) and a
training step at batch 6, next to fp32."""
import sys
import time
import torch
sys.path.insert(0, ".")
from erfnet_pytorch_amd import erfnet_nobn, net_classification
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d

DEV = "cuda:0"
torch.manual_seed(0)


def fwd(dtype, h, w, iters=30):
    net = erfnet_nobn.ERFNet(19).to(DEV).eval()
    x = torch.randn(1, 3, h, w, device=DEV).to(dtype)
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            net(x)
        torch.cuda.synchronize()
    print(f"fwd {dtype} {h}x{w}: {(time.time() - t0) / iters * 1e3:.3f} ms/img")


def train(dtype, iters=20):
    net = net_classification.Net(softmax_classes=20).to(DEV).train()
    opt = torch.optim.Adam([p for p in net.parameters()], 1e-4, eps=1e-4, weight_decay=1e-6)
    crit = CrossEntropyLoss2d()
    x = torch.rand(6, 3, 512, 1024, device=DEV).to(dtype)
    lab = torch.randint(-1, 20, (6, 512, 1024), device=DEV)
    def step():
        opt.zero_grad(set_to_none=True)
        out, _ = net(x)
        loss = crit(out, lab)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):
        l = step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        l = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    print(f"train {dtype} b6: {dt * 1e3:.2f} ms/step {6 / dt:.1f} img/s loss {l.item():.4f}")


for dt in (torch.float32, torch.bfloat16):
    fwd(dt, 512, 1024)
    fwd(dt, 1024, 2048)
    train(dt)
