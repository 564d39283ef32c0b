"""Fixed vs per-K-step cost of the igemm kernel: 1x1 (1 tap) and 3x3 (9 taps) convs, same pixels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf
dev = "cuda:0"
def bench(c, n, h, w, k, reps=20):
    x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, k, k, device=dev) * 0.05
    b = torch.zeros(c, device=dev)
    with torch.no_grad():
        for _ in range(3): hipf.conv2d(x, wt, b)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): hipf.conv2d(x, wt, b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    fl = 2.0 * n * h * w * k * k * c * c
    print(f"C={c} {n}x{h}x{w} k={k}: {dt*1e6:7.1f} us  {fl/dt/1e12:6.1f} TF")
    return dt
for c, n, h, w in ((128, 6, 64, 128), (64, 6, 128, 256), (128, 1, 64, 128)):
    t1 = bench(c, n, h, w, 1); t9 = bench(c, n, h, w, 3)
    per_tap = (t9 - t1) / 8
    print(f"   per-tap {per_tap*1e6:.1f} us, fixed {(t1 - per_tap)*1e6:.1f} us (includes pack launch)")
