"""One workload for rocprofv3: `prof_step.py train|fwd f32|bf16 [H W]`."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet_nobn, net_classification, optim
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d

DEV = "cuda:0"
what, dts = sys.argv[1], sys.argv[2]
dtype = torch.bfloat16 if dts == "bf16" else torch.float32
torch.manual_seed(0)
if what == "fwd":
    h, w = int(sys.argv[3]), int(sys.argv[4])
    net = erfnet_nobn.ERFNet(19).to(DEV).eval()
    x = torch.randn(1, 3, h, w, device=DEV).to(dtype)
    with torch.no_grad():
        for _ in range(20):
            net(x)
else:
    net = net_classification.Net(softmax_classes=20).to(DEV).train()
    opt = optim.Adam([p for p in net.parameters()], 1e-4, eps=1e-4, weight_decay=1e-6)
    crit = CrossEntropyLoss2d()
    x = torch.rand(6, 3, 512, 1024, device=DEV).to(dtype)
    lab = torch.randint(-1, 20, (6, 512, 1024), device=DEV)
    for _ in range(10):
        opt.zero_grad(set_to_none=True)
        out, _ = net(x)
        crit(out, lab).backward()
        opt.step()
torch.cuda.synchronize()
