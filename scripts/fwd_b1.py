"""Batch-1 forward of erfnet_nobn.ERFNet(19) at 512x1024 (the forward-time harness' workload), a few
eager iterations for rocprofv3 kernel traces."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import erfnet_nobn
m = erfnet_nobn.ERFNet(19).to("cuda:0").eval()
x = torch.randn(int(os.environ.get("B", "1")), 3, 512, 1024, device="cuda:0")
with torch.no_grad():
    for _ in range(6):
        m(x)
torch.cuda.synchronize()
