"""Micro-benchmark: repeated non_bottleneck_1d forward (inference form = 4 igemm launches) or
forward+backward at one shape, for rocprofv3 kernel-trace / PMC runs."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=128); ap.add_argument("--h", type=int, default=64)
ap.add_argument("--w", type=int, default=128); ap.add_argument("--n", type=int, default=6)
ap.add_argument("--d", type=int, default=2); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--bwd", action="store_true"); ap.add_argument("--bf16", action="store_true"); ap.add_argument("--copy", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(a.n, a.c, a.h, a.w, generator=g).clamp_(min=0).to(dev).contiguous(memory_format=torch.channels_last)
if a.bf16: x = x.to(torch.bfloat16)
ws = [(torch.randn(a.c, a.c, *k, generator=g) * 0.05).to(dev).requires_grad_(a.bwd) for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
bs = [torch.zeros(a.c, device=dev).requires_grad_(a.bwd) for _ in range(4)]
gy = torch.randn(a.n, a.c, a.h, a.w, generator=g).to(dev).contiguous(memory_format=torch.channels_last).to(x.dtype)
ybuf = torch.empty_like(x)
def run():
    if a.copy:          # reference point: four plain device copies of the same tensor
        for _ in range(4): ybuf.copy_(x)
    elif a.bwd:
        xx = x.clone().requires_grad_(True)
        y = hipf.nb1d(xx, ws, bs, a.d); y.backward(gy)
    else:
        with torch.no_grad(): hipf.nb1d(x, ws, bs, a.d)
for _ in range(2): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(a.reps): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / a.reps
fl = 4 * 2.0 * a.n * a.h * a.w * 3 * a.c * a.c * (3 if a.bwd else 1)
print(f"C={a.c} {a.n}x{a.h}x{a.w} d={a.d} bwd={a.bwd}: {dt*1e6:.1f} us/iter  {fl/dt/1e12:.1f} TFLOP/s")
