"""GPU parity tests of the split-bf16 path for fp32 tensors (ERF_OPT_SPLIT_BF16): the 64/128-channel
1-D convolutions and their weight gradients computed on the bf16 matrix pipe from fp32 operands
split into three bf16 terms.  The claim tested is fp32-level accuracy: the error against a float64
evaluation must not exceed that of the fp32-MFMA kernels by more than a small factor, and the
oracle comparisons of the fp32 path must pass unchanged with the option on."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_hip_parity as thp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def split_on():
    from erfnet_pytorch_amd import _lib
    prev = _lib.set_option(_lib.OPT_SPLIT_BF16, 2)      # 2 = every eligible convolution, whatever its size
    yield
    _lib.set_option(_lib.OPT_SPLIT_BF16, prev)


@pytest.mark.parametrize("C_,dil,N,H,W", [(64, 4, 2, 20, 24), (128, 16, 1, 20, 24), (128, 2, 2, 70, 136),
                                          (64, 1, 1, 130, 260)])
def test_nb1d_c_abi_against_oracle_split(split_on, C_, dil, N, H, W):
    """erf_nb1d_forward / erf_nb1d_backward vs the CPU oracle (same masks on both sides), with the
    1-D convolutions and weight gradients on the split-bf16 kernels; includes shapes with several
    tiles per block and ragged line segments."""
    thp.test_nb1d_c_abi_against_oracle(C_, dil, N, H, W)


@pytest.mark.parametrize("C_,dil,N,H,W,mode,drop", [(64, 2, 2, 20, 24, 2, False), (128, 4, 2, 12, 20, 2, True),
                                                    (64, 8, 1, 20, 24, 1, True)])
def test_nb1d_c_abi_batchnorm_against_oracle_split(split_on, C_, dil, N, H, W, mode, drop):
    """BatchNorm variants with the split kernels; at 128 channels in train mode the batch statistics come
    out of the convolution kernel itself."""
    thp.test_nb1d_c_abi_batchnorm_against_oracle(C_, dil, N, H, W, mode, drop)


@pytest.mark.parametrize("C_,axis,dil,N,H,W", [(128, 0, 1, 2, 40, 200), (128, 1, 8, 3, 100, 90), (64, 0, 2, 1, 33, 300),
                                               (64, 1, 1, 2, 260, 70), (128, 1, 16, 1, 9, 17)])
def test_conv_line_split_against_float64(C_, axis, dil, N, H, W):
    """erf_conv_line_forward_split (one convolution, residual + ReLU + mask epilogue) against torch's
    float64 convolution on the CPU; tolerance = a few fp32 roundings of the accumulated sum."""
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator().manual_seed(C_ + 7 * axis + dil)
    x = torch.randn(N, C_, H, W, generator=g)
    ks = (3, 1) if axis == 0 else (1, 3)
    w = torch.randn(C_, C_, *ks, generator=g) * 0.05
    b = torch.randn(C_, generator=g) * 0.1
    res = torch.randn(N, C_, H, W, generator=g)
    mask = torch.randn(N, C_, H, W, generator=g)
    pad = (dil, 0) if axis == 0 else (0, dil)
    dl = (dil, 1) if axis == 0 else (1, dil)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad, dilation=dl) + res.double())
    ref = torch.where(mask.double() > 0, ref, torch.zeros_like(ref))
    packed = hipf.conv_line_pack_split(w.to(DEV))
    y = hipf.conv_line_split(x.to(DEV), packed, b.to(DEV), axis, dil, True, residual=res.to(DEV), mask=mask.to(DEV))
    err = (y.double().cpu() - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() <= 4e-6 * scale, (err.max().item(), scale)
    # and without the optional tensors
    ref2 = F.conv2d(x.double(), w.double(), b.double(), padding=pad, dilation=dl)
    y2 = hipf.conv_line_split(x.to(DEV), packed, b.to(DEV), axis, dil, False)
    err2 = (y2.double().cpu() - ref2).abs()
    assert err2.max().item() <= 4e-6 * ref2.abs().max().item()


def test_split_path_is_as_accurate_as_fp32_mfma():
    """Whole block forward at a training shape: rms error against float64 of the split path within
    1.5x of the fp32-MFMA path's, and the two paths really are different kernels."""
    from erfnet_pytorch_amd import _lib, erfnet_blocks
    torch.manual_seed(5)
    blk = erfnet_blocks.non_bottleneck_1d(128, 0.0, 2).to(DEV).eval()
    x = torch.randn(2, 128, 64, 128, device=DEV).relu_().contiguous(memory_format=torch.channels_last)
    xd = x.double().cpu()
    o = xd
    for i, name in enumerate(("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")):
        m = getattr(blk, name)
        o = F.conv2d(o, m.weight.double().cpu(), m.bias.double().cpu(), padding=m.padding, dilation=m.dilation)
        o = F.relu(o + xd) if i == 3 else F.relu(o)
    ys = []
    for opt in (0, 2):
        prev = _lib.set_option(_lib.OPT_SPLIT_BF16, opt)
        try:
            with torch.no_grad():
                ys.append(blk(x).double().cpu())
        finally:
            _lib.set_option(_lib.OPT_SPLIT_BF16, prev)
    rms = [(y - o).pow(2).mean().sqrt().item() for y in ys]
    assert not torch.equal(ys[0], ys[1])
    assert rms[1] <= 1.5 * rms[0] + 1e-9, rms
    assert (ys[1] - o).abs().max().item() <= 2e-6 * o.abs().max().item()


def test_batch_statistics_from_the_convolution_kernel(split_on):
    """Train-mode BatchNorm behind a 128-channel split convolution: the kernel leaves the column sums of
    its output itself.  Several ragged tiles per block; the running statistics after one forward must
    equal those of the float64 pre-normalisation tensor (momentum 0.1, unbiased variance)."""
    from erfnet_pytorch_amd import erfnet_blocks
    torch.manual_seed(11)
    blk = erfnet_blocks.non_bottleneck_1d(128, 0.0, 2, use_batchnorm=True).to(DEV).train()
    x = torch.randn(2, 128, 70, 136, device=DEV).relu_().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        blk(x)
    xd = x.double().cpu()
    w = lambda m: (m.weight.detach().double().cpu(), m.bias.detach().double().cpu())
    o = F.relu(F.conv2d(xd, *w(blk.conv3x1_1), padding=blk.conv3x1_1.padding))
    pre = F.conv2d(o, *w(blk.conv1x3_1), padding=blk.conv1x3_1.padding)
    mean = pre.mean(dim=(0, 2, 3))
    var_unb = pre.var(dim=(0, 2, 3), unbiased=True)
    got_mean = blk.bn1.running_mean.double().cpu() / 0.1
    got_var = (blk.bn1.running_var.double().cpu() - 0.9) / 0.1
    assert (got_mean - mean).abs().max().item() <= 2e-6 * max(1.0, mean.abs().max().item())
    # (running_var is an fp32 number near 0.9 + 0.1 var: recovering var from it costs ~1e-6 absolute)
    assert ((got_var - var_unb).abs() - 2e-5 * var_unb).max().item() <= 3e-6
    assert int(blk.bn1.num_batches_tracked) == 1


def test_split_kernels_on_non_finite_and_tiny_inputs():
    """Documented edge behaviour of the split arithmetic (x = hi + mid + lo in bf16):
      * a non-finite input element makes every output that reads it NaN (inf - inf in the split),
        where an fp32 convolution would give +-inf or NaN; finite neighbours are unaffected;
      * magnitudes down to 2^-100 keep all three terms (the low term is 2^-16 of the value, far above
        the smallest bf16 normal 2^-126): relative error stays at the fp32 level;
      * below about 2^-110 the low terms underflow in bf16 and the result degrades towards the
        precision of the surviving terms — it must still be within 2^-8 relative (the high term alone)
        and exactly zero inputs give exactly zero."""
    from erfnet_pytorch_amd import functional as hipf
    C_, N, H, W, dil = 64, 1, 24, 70, 1
    g = torch.Generator().manual_seed(123)
    w = torch.randn(C_, C_, 3, 1, generator=g) * 0.05
    b = torch.zeros(C_)
    packed = hipf.conv_line_pack_split(w.to(DEV))
    x = torch.randn(N, C_, H, W, generator=g)

    def run(xx):
        return hipf.conv_line_split(xx.to(DEV), packed, b.to(DEV), 0, dil, False).cpu()

    def ref64(xx):
        return F.conv2d(xx.double(), w.double(), None, padding=(dil, 0), dilation=(dil, 1))

    # non-finite
    for bad in (float("inf"), float("-inf"), float("nan")):
        xb = x.clone()
        xb[0, 5, 10, 33] = bad
        y = run(xb)
        hit = torch.zeros(N, 1, H, W, dtype=torch.bool)
        hit[0, 0, 9:12, 33] = True                       # the three output rows whose taps read row 10
        assert torch.isnan(y[hit.expand_as(y)]).all(), bad
        clean = ~hit.expand_as(y)
        r = ref64(x)
        assert torch.isfinite(y[clean]).all()
        assert (y.double() - r)[clean].abs().max() <= 4e-6 * r.abs().max()
    # tiny magnitudes: full accuracy at 2^-100
    for e, tol in ((-100, 4e-6), (-120, 2.0 ** -8)):
        xs = x * 2.0 ** e
        y, r = run(xs), ref64(xs)
        assert torch.isfinite(y).all()
        assert (y.double() - r).abs().max() <= tol * r.abs().max(), e
    assert torch.equal(run(torch.zeros_like(x)), torch.zeros(N, C_, H, W))


def _pack_sign_bits(t):
    """(t > 0) of an [N,C,H,W] tensor as the library lays it out: word [pixel][c / 32], bit c % 32."""
    n, c, h, w = t.shape
    b = (t.permute(0, 2, 3, 1).reshape(-1, c // 32, 32) > 0).to(torch.int64)
    return (b << torch.arange(32, device=t.device)).sum(dim=2).reshape(-1)


def _expected_bits_fp32(C_, H, W, dils, bn_mode):
    """Rolling kernel: a convolution along H takes the image rows as lines (H of them, W pixels long),
    one along W the columns; it wants 64-pixel tiles (128 channels: lines longer than 32 pixels) and
    line counts that divide by the dilation.  Behind a BatchNorm the ReLU is the normalisation pass's."""
    out = []
    for d in dils:
        geo = [(H, W, 1), (W, H, 1), (H, W, d), (W, H, d)]
        fits = [(C_ == 64 or length > 32) and lines % dl == 0 for lines, length, dl in geo]
        # (behind a BatchNorm the ReLU belongs to the normalisation pass, which always writes its bits)
        out.append(sum(1 << k for k in range(4) if fits[k] or (bn_mode and k in (1, 3))))
    return out


def _expected_bits_bf16(C_, H, W, dils, bn_mode):
    """Eight-wave pair kernel: conv1x3_1 + conv3x1_2 write t2 and t3, conv1x3_2 + the next block's
    conv3x1_1 write y and that block's t1 (no BatchNorm, at least two columns per block of the grid)."""
    nb = len(dils)
    return [6 | (8 if i + 1 < nb else 0) | (1 if i > 0 else 0) for i in range(nb)]


def _run_bits_case(dtype, C_, N, H, W, dils, bn_mode, expected):
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator(device="cpu").manual_seed(C_ + H)
    x = torch.randn(N, C_, H, W, generator=g).clamp_(min=0).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, C_, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    blocks = []
    for d in dils:
        ws = [(torch.randn(C_, C_, *k, generator=g) * (0.5 / C_ ** 0.5)).to(DEV).requires_grad_(True)
              for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
        bs = [(torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True) for _ in range(4)]
        b = dict(weights=ws, biases=bs, dilated=d)
        if bn_mode:
            b["bn_mode"] = bn_mode
            for key in ("bn1", "bn2"):
                b[key] = ((torch.rand(C_, generator=g) + 0.5).to(DEV).requires_grad_(True),
                          (torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True),
                          torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV))
        blocks.append(b)
    leaves = [q for b in blocks for q in b["weights"] + b["biases"] +
              ([b["bn1"][0], b["bn1"][1], b["bn2"][0], b["bn2"][1]] if bn_mode else [])]
    res = []
    for on in (False, True):
        prev = hipf.set_sign_bits(on)
        try:
            for q in leaves:
                q.grad = None
            xx = x.clone().requires_grad_(True)
            y = hipf.nb1d_chain(xx, blocks)
            node = y.grad_fn
            while node is not None and not hasattr(node, "bits_written"):
                node = node.next_functions[0][0] if node.next_functions else None
            assert node is not None
            if on:
                per = 18
                saved = node.saved_tensors
                assert list(node.bits_written) == expected, (node.bits_written, expected)
                for i in range(len(dils)):
                    blk = saved[1 + i * per:1 + (i + 1) * per]
                    acts = {0: blk[1], 1: blk[3], 2: blk[4], 3: blk[0]}        # t1, t2, t3, y
                    for k in range(4):
                        if node.bits_written[i] >> k & 1:
                            got = blk[17][k].to(torch.int64) & 0xFFFFFFFF
                            assert torch.equal(got, _pack_sign_bits(acts[k])), (i, k)
            else:
                assert not any(node.bits_written)
            y.backward(gy)
            res.append([y.detach(), xx.grad] + [q.grad.clone() for q in leaves])
        finally:
            hipf.set_sign_bits(prev)
    for i, (u, v) in enumerate(zip(*res)):
        assert torch.equal(u, v), f"tensor {i}"


@pytest.mark.parametrize("C_,N,H,W,dils,bn_mode", [(64, 2, 40, 72, (1, 2, 4), 0), (128, 1, 33, 64, (2, 16, 1), 0),
                                                   (64, 1, 24, 100, (1, 8), 2), (128, 2, 20, 24, (4, 2), 2),
                                                   (128, 1, 48, 40, (4, 2), 2), (64, 1, 20, 30, (3, 1), 0)])
def test_relu_masks_as_bits(split_on, C_, N, H, W, dils, bn_mode):
    """A run of blocks keeps its ReLU masks as bit tensors (erf_nb1d_block_t::sign_bits) and hands them to
    erf_nb1d_backward_bits: the bits written equal (activation > 0), every launch that can write or
    read them does (without BatchNorm all four tensors of a block when the lines divide by the
    dilation, with it t1 and t3), and every gradient equals the run with the masks read from the
    activations bit for bit."""
    _run_bits_case(torch.float32, C_, N, H, W, dils, bn_mode, _expected_bits_fp32(C_, H, W, dils, bn_mode))


@pytest.mark.parametrize("C_,N,H,W,dils", [(64, 4, 32, 128, (1, 2, 4)), (128, 4, 24, 128, (2, 4)), (64, 5, 100, 104, (8,)),
                                           (128, 9, 64, 60, (16, 1, 2, 3))])
def test_relu_masks_as_bits_bf16(C_, N, H, W, dils):
    """The same on bf16 tensors, where the paired launches write and read the bits (ragged tiles and
    single-block runs included)."""
    _run_bits_case(torch.bfloat16, C_, N, H, W, dils, 0, _expected_bits_bf16(C_, H, W, dils, 0))


@pytest.mark.parametrize("kind,cin,cout,N,H,W", [("down", 16, 64, 2, 40, 72), ("down", 64, 128, 3, 34, 50),
                                                 ("up", 128, 64, 2, 17, 40), ("conv3x3", 64, 20, 1, 20, 36)])
def test_generic_convolutions_split_against_float64(kind, cin, cout, N, H, W):
    """The stride-2 layers (and the other generic convolutions) of fp32 tensors on igemm_x3_kernel: the output
    against float64 within 4e-6 of its scale and no less accurate than the fp32-MFMA kernel (split arithmetic
    switched off) by more than a small factor; the input gradient within 1e-5 relative L2."""
    from erfnet_pytorch_amd import _lib, functional as hipf
    import golden_util as gu
    wshape = {"down": (cout - cin, cin, 3, 3), "up": (cin, cout, 3, 3), "conv3x3": (cout, cin, 3, 3)}[kind]
    nb = wshape[1] if kind == "up" else wshape[0]
    w = torch.from_numpy(gu.det_normal("gx3.w." + kind, wshape) * (0.5 / np.sqrt(9 * cin))).to(DEV)
    b = torch.from_numpy(0.1 * gu.det_normal("gx3.b." + kind, (nb,))).to(DEV)
    x = torch.from_numpy(np.maximum(gu.det_normal("gx3.x." + kind, (N, cin, H, W)), 0)).to(DEV)

    def composite(xx, ww, bb, hip):
        if kind == "down":
            return hipf.downsampler(xx, ww, bb) if hip else torch.relu(torch.cat(
                [F.conv2d(xx, ww, bb, stride=2, padding=1), F.max_pool2d(xx, 2, 2)], 1))
        if kind == "up":
            return hipf.upsampler(xx, ww, bb) if hip else torch.relu(
                F.conv_transpose2d(xx, ww, bb, stride=2, padding=1, output_padding=1))
        return hipf.conv2d(xx, ww, bb) if hip else F.conv2d(xx, ww, bb, padding=1)

    x64 = x.double().requires_grad_(True)
    ref = composite(x64, w.double(), b.double(), False)
    gy = torch.from_numpy(gu.det_normal("gx3.gy." + kind, tuple(ref.shape))).to(DEV)
    ref.backward(gy.double())
    scale = ref.detach().abs().max().item()

    def run():
        xx = x.clone().requires_grad_(True)
        y = composite(xx, w, b, True)
        y.backward(gy)
        e_y = (y.detach().double() - ref.detach()).abs().max().item() / scale
        e_g = ((xx.grad.double() - x64.grad).norm() / x64.grad.norm()).item()
        return e_y, e_g

    e_split, g_split = run()
    prev = _lib.set_option(_lib.OPT_SPLIT_BF16, 0)
    try:
        e_fp32, g_fp32 = run()
    finally:
        _lib.set_option(_lib.OPT_SPLIT_BF16, prev)
    assert e_split <= 4e-6, (e_split, e_fp32)
    assert e_split <= 3 * e_fp32 + 1e-7, (e_split, e_fp32)
    assert g_split <= 1e-5, (g_split, g_fp32)
