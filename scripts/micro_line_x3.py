"""Micro-benchmark of line_x3_kernel alone (one 3-tap convolution on fp32 tensors, split-bf16
arithmetic): weights packed once, then `reps` launches — for rocprofv3 kernel-trace / PMC passes and
HIP-event timing."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=128); ap.add_argument("--h", type=int, default=64)
ap.add_argument("--w", type=int, default=128); ap.add_argument("--n", type=int, default=6)
ap.add_argument("--d", type=int, default=2); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--axis", type=int, default=0); ap.add_argument("--res", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(a.n, a.c, a.h, a.w, generator=g).clamp_(min=0).to(dev).contiguous(memory_format=torch.channels_last)
k = (3, 1) if a.axis == 0 else (1, 3)
w = (torch.randn(a.c, a.c, *k, generator=g) * 0.05).to(dev)
b = torch.zeros(a.c, device=dev)
res = x.clone() if a.res else None
packed = hipf.conv_line_pack_split(w)
for _ in range(3):
    hipf.conv_line_split(x, packed, b, a.axis, a.d, True, residual=res)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.reps):
    hipf.conv_line_split(x, packed, b, a.axis, a.d, True, residual=res)
e.record(); e.synchronize()
us = s.elapsed_time(e) * 1e3 / a.reps
fl = 2.0 * a.n * a.h * a.w * 3 * a.c * a.c
print(f"line_x3 C={a.c} {a.n}x{a.h}x{a.w} d={a.d} axis={a.axis} res={a.res}: {us:.1f} us/launch  {fl/us/1e6:.1f} TFLOP/s (algorithmic)")
