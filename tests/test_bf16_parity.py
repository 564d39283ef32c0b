"""bf16 storage path (bf16 activations, fp32 parameters, fp32 accumulation) against fp32/fp64
references on the same bf16-rounded inputs.  Tolerance: the north star's 1e-2 for bf16, taken
relative to the tensor's scale (max |reference|), plus the rounding of the stored result."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rb(t):
    """round to bf16 and back (fp64 carrier)"""
    return t.to(BF).double()


def _check(got, ref, what, tol=1e-2):
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def _conv1d(x, w, b, axis, dil):
    """x [N,C,H,W] fp64; w [C,C,3]; axis 0 = along H (3x1), 1 = along W (1x3)"""
    if axis == 0:
        return F.conv2d(x, w[:, :, :, None], b, padding=(dil, 0), dilation=(dil, 1))
    return F.conv2d(x, w[:, :, None, :], b, padding=(0, dil), dilation=(1, dil))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("c,h,w,first_axis,d1,d2,pair,res", [
    (128, 40, 56, 1, 1, 2, True, True),      # columns of 40 pixels: tile 64, ragged
    (128, 64, 32, 1, 2, 16, True, False),
    (128, 24, 40, 0, 4, 1, True, True),      # rows as lines
    (64, 40, 56, 1, 1, 1, True, True),
    (64, 128, 48, 1, 1, 3, True, False),
    (64, 200, 24, 1, 2, 2, True, True),      # tile 256
    (128, 100, 36, 1, 8, 8, True, True),     # tile 128
    (128, 40, 56, 0, 2, 0, False, True),     # single convolution along H, rows of 56
    (128, 24, 300, 0, 1, 0, False, False),   # single, rows of 300 pixels: three segments
    (64, 30, 600, 0, 1, 0, False, True),     # single, 64 channels, three segments of 256
    (64, 70, 20, 1, 5, 0, False, False),     # single along W
])
def test_line_kernel_against_fp64(c, h, w, first_axis, d1, d2, pair, res):
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    n = 2
    tag = f"bf16line.{c}.{h}.{w}.{first_axis}.{d1}.{d2}"
    x = torch.as_tensor(gu.det_normal(tag + ".x", (n, c, h, w)))
    w1 = torch.as_tensor(gu.det_normal(tag + ".w1", (c, c, 3))) * (1.0 / (3 * c) ** 0.5)
    w2 = torch.as_tensor(gu.det_normal(tag + ".w2", (c, c, 3))) * (1.0 / (3 * c) ** 0.5)
    b1 = torch.as_tensor(gu.det_normal(tag + ".b1", (c,))) * 0.1
    b2 = torch.as_tensor(gu.det_normal(tag + ".b2", (c,))) * 0.1
    r = torch.as_tensor(gu.det_normal(tag + ".r", (n, c, h, w))) if res else None
    # reference on the bf16-rounded operands, with the kernel's rounding points
    xr, w1r, w2r = _rb(x), _rb(w1), _rb(w2)
    mid = torch.relu(_conv1d(xr, w1r, b1.double(), first_axis, d1))
    if pair:
        mid = _rb(mid)
        out = _conv1d(mid, w2r, b2.double(), 1 - first_axis, d2)
    else:
        out = _conv1d(xr, w1r, b1.double(), first_axis, d1)
    if res:
        out = out + _rb(r)
    out = torch.relu(out)

    xd = _nhwc(x.to(DEV).to(BF))
    rd = _nhwc(r.to(DEV).to(BF)) if res else None
    w1d, w2d, b1d, b2d = (t.to(DEV).float().contiguous() for t in (w1, w2, b1, b2))
    packed = torch.empty(2 * 3 * c * c, dtype=BF, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.erf_conv_pair_pack_bf16(_ptr(w1d), _ptr(w2d) if pair else None, c, _ptr(packed), st))
    y = torch.full_like(xd, 7.0)
    m = torch.full_like(xd, 7.0) if pair else None
    before = _lib.launch_count()
    _lib.check(lib.erf_conv_pair_forward_bf16(_ptr(xd), _ptr(packed), _ptr(b1d), _ptr(b2d), first_axis, d1, d2,
                                              1, 1, _ptr(rd), _ptr(m), _ptr(y), n, c, h, w, st))
    torch.cuda.synchronize()
    assert _lib.launch_count() == before + 1
    if pair:
        _check(m, mid, "mid", tol=2.0 ** -7)
    _check(y, out, "out", tol=2.0 ** -7)
