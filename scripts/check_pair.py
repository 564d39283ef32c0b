"""Fused convolution pairs vs. one launch per convolution: bitwise comparison and timing of a
non_bottleneck_1d (inference forward, training forward + backward) at a few shapes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import functional as hipf, _lib

dev = "cuda:0"


def case(n, c, h, w, d, reps=10):
    g = torch.Generator().manual_seed(c + h + d)
    x = torch.randn(n, c, h, w, generator=g).clamp_(min=0).to(dev).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(c, c, *k, generator=g) * 0.05).to(dev).requires_grad_(True) for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
    bs = [(torch.randn(c, generator=g) * 0.1).to(dev).requires_grad_(True) for _ in range(4)]
    gy = torch.randn(n, c, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)

    def infer():
        with torch.no_grad():
            return hipf.nb1d(x, ws, bs, d)

    def train():
        xx = x.clone().requires_grad_(True)
        for t in ws + bs:
            t.grad = None
        y = hipf.nb1d(xx, ws, bs, d)
        y.backward(gy)
        return [y.detach(), xx.grad] + [t.grad for t in ws + bs]

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e6

    res = {}
    for on in (0, 1):
        _lib.set_option(_lib.OPT_PAIR_FUSION, on)
        k0 = _lib.launch_count(); yi = infer(); ki = _lib.launch_count() - k0
        k0 = _lib.launch_count(); tr = train(); kt = _lib.launch_count() - k0
        res[on] = (yi, tr, ki, kt, timed(infer), timed(train))
    same_i = torch.equal(res[0][0], res[1][0])
    same_t = all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    md = max((a - b).abs().max().item() for a, b in zip(res[0][1], res[1][1]))
    print(f"C={c} {n}x{h}x{w} d={d}: launches infer {res[0][2]}->{res[1][2]} train {res[0][3]}->{res[1][3]}  "
          f"bitwise infer={same_i} train={same_t} (max diff {md:.2e})  "
          f"infer {res[0][4]:.1f}->{res[1][4]:.1f} us  train {res[0][5]:.1f}->{res[1][5]:.1f} us", flush=True)
    return same_i and same_t


ok = True
for shape in [(4, 128, 28, 28, 2), (4, 128, 28, 28, 16), (4, 64, 56, 56, 1), (3, 128, 17, 40, 4), (2, 128, 40, 72, 2), (2, 64, 72, 40, 1), (1, 128, 64, 24, 16), (6, 128, 64, 128, 2), (6, 128, 64, 128, 16),
              (6, 64, 128, 256, 1)]:
    ok = case(*shape) and ok
print("ALL BITWISE EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
