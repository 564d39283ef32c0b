"""Micro-benchmark of the fused convolution pair (pair_f32_kernel) alone: weights packed once, then
`reps` launches — for rocprofv3 kernel-trace / PMC passes and HIP-event timing."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=128); ap.add_argument("--h", type=int, default=64)
ap.add_argument("--w", type=int, default=128); ap.add_argument("--n", type=int, default=6)
ap.add_argument("--d", type=int, default=2); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--axis", type=int, default=1); ap.add_argument("--no-mid", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(a.n, a.c, a.h, a.w, generator=g).clamp_(min=0).to(dev).contiguous(memory_format=torch.channels_last)
ka, kb = ((1, 3), (3, 1)) if a.axis == 1 else ((3, 1), (1, 3))
wa = (torch.randn(a.c, a.c, *ka, generator=g) * 0.05).to(dev)
wb = (torch.randn(a.c, a.c, *kb, generator=g) * 0.05).to(dev)
b = torch.zeros(a.c, device=dev)
packed = hipf.conv_pair_pack(wa, wb)
for _ in range(3):
    hipf.conv_pair(x, packed, b, b, a.axis, a.d, a.d, keep_mid=not a.no_mid)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.reps):
    hipf.conv_pair(x, packed, b, b, a.axis, a.d, a.d, keep_mid=not a.no_mid)
e.record(); e.synchronize()
us = s.elapsed_time(e) * 1e3 / a.reps
fl = 2 * 2.0 * a.n * a.h * a.w * 3 * a.c * a.c
print(f"pair C={a.c} {a.n}x{a.h}x{a.w} d={a.d} axis={a.axis} mid={not a.no_mid}: {us:.1f} us/launch  {fl/us/1e6:.1f} TFLOP/s")
