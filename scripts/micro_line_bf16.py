"""Times the bf16 line kernel (pair / single) and the fp32 pair kernel on the shapes of the network."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from erfnet_pytorch_amd import _lib

lib = _lib.lib()
DEV = "cuda:0"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def run(n, c, h, w, first_axis, d1, d2, keep_mid, res, iters=50):
    x = torch.randn(n, c, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(x) if res else None
    w1 = torch.randn(c, c, 3, device=DEV) * 0.05
    w2 = torch.randn(c, c, 3, device=DEV) * 0.05
    b = torch.zeros(c, device=DEV)
    packed = torch.empty(2 * 3 * c * c, dtype=torch.bfloat16, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.erf_conv_pair_pack_bf16(p(w1), p(w2) if d2 else None, c, p(packed), st))
    y = torch.empty_like(x)
    m = torch.empty_like(x) if keep_mid else None
    def call():
        _lib.check(lib.erf_conv_pair_forward_bf16(p(x), p(packed), p(b), p(b), first_axis, d1, d2, 1, 1, p(r), p(m), p(y),
                                                  n, c, h, w, st))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    px = n * h * w
    nconv = 2 if d2 else 1
    flop = nconv * 2 * 3 * c * c * px
    byts = px * c * 2 * (2 + (1 if keep_mid else 0) + (1 if res else 0))
    print(f"N{n} C{c} {h}x{w} axis{first_axis} d{d1},{d2} mid{int(keep_mid)} res{int(res)}: {us:7.1f} us  "
          f"{flop / us * 1e-6:7.1f} TFLOP/s  {byts / us * 1e-3:7.1f} GB/s alg")


CASES_IMAGENET = [
    (256, 128, 28, 28, 1, 1, 2, True, False),
    (256, 128, 28, 28, 1, 1, 2, True, True),
    (256, 128, 28, 28, 0, 1, 0, False, False),
    (256, 64, 56, 56, 1, 1, 1, True, False),
    (32, 128, 28, 28, 1, 1, 2, True, False),
    (256, 128, 28, 28, 1, 1, 2, False, False),
    (256, 128, 32, 32, 1, 1, 2, True, False),
    (256, 128, 28, 32, 1, 1, 2, True, False),
    (256, 128, 32, 28, 1, 1, 2, True, False),
    (64, 128, 64, 28, 1, 1, 2, True, False),
    (256, 128, 28, 28, 0, 1, 2, True, False),
    (256, 128, 28, 28, 1, 1, 1, True, False),
]
if len(sys.argv) > 1 and sys.argv[1] == "imagenet":
    for args in CASES_IMAGENET:
        run(*args, iters=10)
    sys.exit(0)

for args in [
    (6, 128, 64, 128, 1, 1, 2, False, True),
    (6, 128, 64, 128, 1, 1, 2, True, True),
    (6, 128, 64, 128, 0, 1, 0, False, False),
    (6, 64, 128, 256, 1, 1, 1, False, True),
    (6, 64, 128, 256, 1, 1, 1, True, True),
    (6, 64, 128, 256, 0, 1, 0, False, False),
    (1, 128, 128, 256, 1, 1, 2, False, True),
    (1, 128, 128, 256, 0, 1, 0, False, False),
    (1, 64, 256, 512, 1, 1, 1, False, True),
    (1, 64, 256, 512, 0, 1, 0, False, False),
    (1, 128, 64, 128, 1, 1, 2, False, True),
]:
    run(*args)
