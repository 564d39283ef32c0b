"""Forward of one non_bottleneck_1d at the two training shapes, split-bf16 path on (run under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import _lib, erfnet_blocks
_lib.lib().erf_set_option(_lib.OPT_SPLIT_BF16, 2)
for (c, d, shape) in ((128, 2, (6, 128, 64, 128)), (64, 1, (6, 64, 128, 256))):
    blk = erfnet_blocks.non_bottleneck_1d(c, 0.0, d).cuda().eval()
    x = torch.randn(shape, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(30): blk(x)
    torch.cuda.synchronize()
print("done")
