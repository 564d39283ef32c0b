"""Diagnostic: whole-net BN-train backward, HIP fp32 vs CPU float64 / float32 (same module code)."""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import golden_util as gu
from erfnet_pytorch_amd import erfnet

key = "eval_erfnet_20_train"
NETS = gu.load_npz("nets.npz")
def make():
    net = gu.fill_state_dict(erfnet.ERFNet(20), tag=key)
    for sub in net.modules():
        if isinstance(sub, torch.nn.Dropout2d): sub.p = 0
    return net.train()
x = torch.from_numpy(gu.det_uniform("net.x_train", (2, 3, 64, 128), 0.0, 1.0))
lab = torch.from_numpy(gu.det_labels("net.lab_train", (2, 64, 128), -1, 20))
def run(net, x, lab):
    out = net(x); loss = torch.nn.CrossEntropyLoss(ignore_index=-1)(out, lab); loss.backward()
    return {k: p.grad.detach().double().cpu().numpy() for k, p in net.named_parameters() if p.grad is not None}, loss.item()
g64, l64 = run(make().double(), x.double(), lab)
g32, l32 = run(make(), x, lab)
ghip, lhip = run(make().cuda(), x.cuda(), lab.cuda())
print("loss", l64, l32, lhip)
def rel(a, b): return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
for k in g64:
    print(f"{k:55s} cpu32 {rel(g32[k], g64[k]):.2e}  hip {rel(ghip[k], g64[k]):.2e}  hip-vs-cpu32 {rel(ghip[k], g32[k]):.2e}")
