"""Where conv16_rows_kernel<float> differs from conv16_kernel<float>: the saved activations of one
16-channel block (t1, t2, t3, y: each the output of one more convolution), several runs."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import _lib, functional as hipf
from erfnet_pytorch_amd.erfnet_blocks import non_bottleneck_1d
DEV = "cuda:0"
torch.manual_seed(3)
n, h, w = 1, 512, 1024
x = torch.randn(n, 16, h, w, device=DEV).relu_().contiguous(memory_format=torch.channels_last)
blk = non_bottleneck_1d(16, 0.0, 1).to(DEV).train()
def run(rows):
    prev = _lib.set_option(_lib.OPT_CONV16_ROWS, rows)
    try:
        xx = x.clone().requires_grad_(True)
        y = blk(xx)
        node = y.grad_fn
        while node is not None and not hasattr(node, "saved_tensors"):
            node = node.next_functions[0][0]
        sv = node.saved_tensors
        torch.cuda.synchronize()
        return [t.clone() for t in sv[:8] if t is not None and t.dim() == 4]
    finally:
        _lib.set_option(_lib.OPT_CONV16_ROWS, prev)
ref = run(0)
print("saved 4-d tensors:", len(ref))
for trial in range(3):
    got = run(1)
    for k, (a, b) in enumerate(zip(ref, got)):
        bad = (a != b)
        nb = int(bad.sum())
        msg = f"trial {trial} tensor {k}: {nb} differ"
        if nb:
            bp = bad.permute(0, 2, 3, 1).reshape(-1, 16)
            px = bp.any(dim=1).nonzero().flatten().cpu().numpy()
            ch = bp.sum(dim=0).cpu().numpy().tolist()
            col, row = px % w, (px // w) % h
            msg += f"; pixels {len(px)} col%16 {np.bincount(col % 16, minlength=16).tolist()} tiles%8 {np.bincount((col // 16) % 8, minlength=8).tolist()} rows {sorted(set(row.tolist()))[:8]}.. per-channel {ch} max {float((a-b).abs().max()):.3g}"
        print(msg)
