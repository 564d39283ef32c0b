"""How long does the host take to ENQUEUE one training step (no sync) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import net_classification
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model = net_classification.Net(softmax_classes=20).to(dev).train()
opt = torch.optim.Adam(model.parameters(), 1e-4, eps=1e-4, weight_decay=1e-6)
crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
x = torch.rand(B, 3, 512, 1024, device=dev); lab = torch.randint(-1, 20, (B, 512, 1024), device=dev)
def step():
    opt.zero_grad(set_to_none=False)
    loss = crit(model(x)[0], lab); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
host = []; total = []
for _ in range(5):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"B={B}: host enqueue {min(host)*1e3:.2f} ms, total {min(total)*1e3:.2f} ms per step")
# forward only
with torch.no_grad():
    for _ in range(2): model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); model(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"   forward: host {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
