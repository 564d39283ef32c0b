"""sha1 over every parameter gradient of one training step on fixed inputs (`grad_hash.py [f32|bf16] [model] [N H W]`):
two runs that print the same digest computed bit-identical gradients (A/B of kernel switches)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet, net_classification  # noqa: E402
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d  # noqa: E402

DEV = "cuda:0"
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
bn = len(sys.argv) > 2 and sys.argv[2] == "erfnet_bn"
torch.manual_seed(0)
net = (erfnet.ERFNet(20) if bn else net_classification.Net(softmax_classes=20)).to(DEV).train()
g = torch.Generator().manual_seed(7)
N, H, W = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (3, 256, 512)
x = torch.rand(N, 3, H, W, generator=g).to(DEV).to(dt)
lab = torch.randint(-1, 20, (N, H, W), generator=g).to(DEV)
out = net(x)
out = out[0] if isinstance(out, tuple) else out
loss = CrossEntropyLoss2d()(out, lab)
loss.backward()
torch.cuda.synchronize()
h = hashlib.sha1()
n = 0
for name, p in net.named_parameters():
    if p.grad is not None:
        h.update(p.grad.detach().cpu().numpy().tobytes())
        n += 1
print("digest", h.hexdigest(), "tensors", n, "loss", float(loss.detach()))
