"""Micro-benchmark: the 64<->16 channel Upsampler / Downsampler blocks at the benchmark's sizes,
forward + backward, for rocprofv3 kernel-trace runs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def nhwc(*s): return torch.randn(*s, generator=g).clamp_(min=0).to(dev).contiguous(memory_format=torch.channels_last)
def run(name, fn, x, params, gy, reps=10):
    def it():
        xx = x.clone().requires_grad_(True)
        fn(xx, *params).backward(gy)
    for _ in range(2): it()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): it()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / reps * 1e6:.1f} us/iter")
n = 6
# Upsampler 64 -> 16: 128x256 -> 256x512
w = (torch.randn(64, 16, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True); b = torch.zeros(16, device=dev, requires_grad=True)
run("up 64->16", hipf.upsampler, nhwc(n, 64, 128, 256), (w, b), nhwc(n, 16, 256, 512))
# Downsampler 16 -> 64: 256x512 -> 128x256
w = (torch.randn(48, 16, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True); b = torch.zeros(48, device=dev, requires_grad=True)
run("down 16->64", hipf.downsampler, nhwc(n, 16, 256, 512), (w, b), nhwc(n, 64, 128, 256))
# Upsampler 128 -> 64 and Downsampler 64 -> 128
w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
run("up 128->64", hipf.upsampler, nhwc(n, 128, 64, 128), (w, b), nhwc(n, 64, 128, 256))
w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
run("down 64->128", hipf.downsampler, nhwc(n, 64, 128, 256), (w, b), nhwc(n, 128, 64, 128))
