import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import _lib, erfnet_blocks
torch.manual_seed(1)
c, d, shape = 128, 1, (1, 128, 8, 64)
blk = erfnet_blocks.non_bottleneck_1d(c, 0.0, d).cuda().eval()
x = torch.randn(shape, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    _lib.lib().erf_set_option(_lib.OPT_SPLIT_BF16, 0); y0 = blk(x).clone()
    _lib.lib().erf_set_option(_lib.OPT_SPLIT_BF16, 2); y1 = blk(x).clone()
e = (y1 - y0).abs()
print("max", e.max().item(), "bad frac", (e > 1e-4).float().mean().item())
bad = (e > 1e-4)
print("bad per channel (first 40):", bad.sum(dim=(0, 2, 3))[:40].tolist())
print("bad per row:", bad.sum(dim=(0, 1, 3)).tolist())
print("bad per col:", bad.sum(dim=(0, 1, 2)).tolist())
