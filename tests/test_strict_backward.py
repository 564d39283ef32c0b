"""Strict backward parity of the blocks around non_bottleneck_1d, through the raw C ABI.

Every backward entry point is fed the ORACLE's saved activations (its output y for the ReLU mask, its
pre-BatchNorm tensor, its batch mean / 1/std), so both sides differentiate the same function at the
same point and every gradient compares element by element:

  * fp32: DownsamplerBlock / UpsamplerBlock (train/erfnet_blocks.py:15-29, 77-88) with BatchNorm
    off / eval / train — dx, dW, db, dgamma, dbeta;
  * bf16 storage: the same two blocks, the output transposed convolution
    (train/net_classification.py:56,64) and non_bottleneck_1d at 16 channels and with BatchNorm.
    Operands are bf16-representable, so the oracle (fp32 on the same values) is the exact function;
    what remains is the rounding of the stored gradient tensors.  Bounds: dx within 2^-6 of its
    scale element-wise (up to three chained bf16 roundings) AND 6e-3 in relative L2; parameter
    gradients (fp32 sums of bf16 products) within 3e-3 of scale and 1.5e-3 in relative L2 without
    BatchNorm.  With BatchNorm the gradient passes one more stored tensor (the BatchNorm backward's
    output) and its per-channel sums cancel strongly, which was measured at 1.5-1.9e-3 relative L2 for
    dW: bounds 1e-2 of scale and 4e-3 relative L2 (non_bottleneck_1d, four chained convolutions:
    6e-3).  One uniform rounding of a tensor is 1.1e-3 relative L2, so all of these are a few
    roundings wide; a term dropped or mis-scaled by three per cent exceeds every one of them.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rb(a):
    """numpy fp32 -> nearest bf16-representable fp32"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(BF).float().numpy()


def _dev_nhwc(a, dtype):
    """numpy NCHW -> device tensor holding NHWC data in `dtype`"""
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).to(DEV).to(dtype)


def _host_nchw(t, shape):
    n, c, h, w = shape
    return t.float().cpu().numpy().reshape(n, h, w, c).transpose(0, 3, 1, 2)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _untie_pool_windows(x):
    """Make the maximum of every 2x2 window unique (bf16 data ties often; which of two equal maxima
    receives the gradient is a convention, not part of the function)."""
    n, c, h, w = x.shape
    v = x.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4).copy()
    for _ in range(8):
        mx = v.max(axis=-1, keepdims=True)
        ties = (v == mx).sum(axis=-1) > 1
        if not ties.any():
            break
        first = np.argmax(v == mx, axis=-1)
        bump = np.zeros_like(v, dtype=bool)
        np.put_along_axis(bump, first[..., None], True, axis=-1)
        bump &= ties[..., None]
        # one bf16 step up (or a small positive value at zero)
        nv = np.where(v > 0, v * (1 + 2.0 ** -7), np.where(v < 0, v * (1 - 2.0 ** -7), 2.0 ** -20))
        v = np.where(bump, _rb(nv.astype(np.float32)), v)
    return v.reshape(n, c, h // 2, w // 2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h, w).astype(np.float32)


def _check(got, ref, what, tol, l2tol, floor=1e-6):
    """`floor`: absolute slack, for gradients that are mathematically zero (the bias in front of a
    train-mode BatchNorm): what is left of them is the cancellation error of a sum over all pixels."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= tol * scale + floor, f"{what}: max err {err:.4g} vs scale {scale:.4g}"
    if scale <= 10 * floor:
        return
    den = np.sqrt((ref * ref).sum())
    if den > 1e-3 * np.sqrt(ref.size):           # (a gradient that is mathematically zero has no relative error)
        l2 = np.sqrt(((got - ref) ** 2).sum()) / den
        assert l2 <= l2tol, f"{what}: relative L2 error {l2:.4g}"


TOL = {  # storage -> (dx max, dx L2, parameter max, parameter L2)
    "f32": (2e-4, 1e-4, 2e-4, 1e-4),
    "bf16": (2.0 ** -6, 6e-3, 3e-3, 1.5e-3),
    "bf16_bn": (2.0 ** -6, 6e-3, 1e-2, 4e-3),
}


def _tol(st, mode):
    return TOL["bf16_bn" if (st == "bf16" and mode) else st]


def _zero_floor(st, gy):
    """rounding left in a per-channel sum of gradient terms that cancel exactly: terms of order one over
    all pixels of a channel, each carrying the storage type's rounding"""
    px = gy.size / gy.shape[1]
    return (2.0 ** -9 if st == "bf16" else 2.0 ** -22) * 16 * np.sqrt(px)


class _Bn:
    """erf_bn_t on device tensors, mirrored by the oracle's BN pack."""

    def __init__(self, tag, c, mode):
        from oracle import oracle as orc
        self.mode, self.c = mode, c
        self.np = dict(gamma=gu.det_uniform(tag + ".g", (c,), 0.5, 1.5), beta=0.1 * gu.det_normal(tag + ".b", (c,)),
                       running_mean=0.1 * gu.det_normal(tag + ".rm", (c,)),
                       running_var=gu.det_uniform(tag + ".rv", (c,), 0.5, 1.5))
        self.orc = orc.BN(mode, **self.np) if mode else None
        self.keep = []

    def struct(self, save_mean=None, save_invstd=None):
        from erfnet_pytorch_amd import _lib
        if not self.mode:
            return None
        d = lambda a: self.keep.append(torch.from_numpy(np.asarray(a, np.float32).copy()).to(DEV)) or self.keep[-1]
        f = {k: d(v) for k, v in self.np.items()}
        sm = d(np.zeros(self.c) if save_mean is None else save_mean)
        si = d(np.zeros(self.c) if save_invstd is None else save_invstd)
        st = _lib.BnStruct(self.mode, f["gamma"].data_ptr(), f["beta"].data_ptr(), f["running_mean"].data_ptr(),
                           f["running_var"].data_ptr(), sm.data_ptr(), si.data_ptr(), 0.1, 1e-3, None)
        self.keep.append(st)
        return C.byref(st)


def _fn(lib, name, st):
    return getattr(lib, name + ("_bf16" if st == "bf16" else ""))


@pytest.mark.parametrize("st", ["f32", "bf16"])
@pytest.mark.parametrize("cin,cout,n,h,w,mode", [(16, 64, 2, 12, 20, 0), (16, 64, 2, 12, 20, 2), (64, 128, 2, 8, 12, 1),
                                                 (64, 128, 1, 12, 16, 2), (3, 16, 2, 16, 24, 2), (3, 16, 1, 16, 24, 0)])
def test_downsampler_backward_strict(st, cin, cout, n, h, w, mode):
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    dt = BF if st == "bf16" else torch.float32
    tag = f"strict_down_{cin}_{cout}_{mode}"
    x = gu.det_normal(tag + ".x", (n, cin, h, w))
    wt = gu.det_normal(tag + ".w", (cout - cin, cin, 3, 3)) * (1.0 / (9 * cin) ** 0.5)
    b = 0.1 * gu.det_normal(tag + ".b", (cout - cin,))
    gy = gu.det_normal(tag + ".gy", (n, cout, h // 2, w // 2))
    if st == "bf16":
        x, wt, gy = _untie_pool_windows(_rb(x)), _rb(wt), _rb(gy)
    bn = _Bn(tag, cout, mode)
    y_ref, pre_ref = orc.downsampler_fwd(x, wt, b, bn.orc)
    dx_ref, dw_ref, db_ref, dg_ref, dbt_ref = orc.downsampler_bwd(x, wt, y_ref, pre_ref, gy, bn.orc)

    xd = _dev_nhwc(x, dt)
    yd, gyd = _dev_nhwc(y_ref, dt), _dev_nhwc(gy, dt)
    pred = None if pre_ref is None else _dev_nhwc(pre_ref, dt)
    wd = torch.from_numpy(wt).to(DEV)
    dx = torch.empty_like(xd)
    dw, db = torch.empty_like(wd), torch.empty(cout - cin, device=DEV)
    dg, dbt = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    wsb = torch.empty(int(lib.erf_workspace_bytes(n * h * w, max(cin, cout))), dtype=torch.uint8, device=DEV)
    bref = bn.struct(None if not mode else bn.orc.save_mean, None if not mode else bn.orc.save_invstd)
    sn, sc, sh, sw = h * w * cin, 1, w * cin, cin
    before = _lib.launch_count()
    _lib.check(_fn(lib, "erf_downsampler_backward", st)(
        _p(xd), sn, sc, sh, sw, _p(wd), bref, _p(yd), _p(pred), _p(gyd), _p(dx), _p(dw), _p(db),
        _p(dg) if mode else None, _p(dbt) if mode else None, n, cin, h, w, cout, 0, _p(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    assert _lib.launch_count() > before
    tx, tx2, tp, tp2 = _tol(st, mode)
    _check(_host_nchw(dx, x.shape), dx_ref, "dx", tx, tx2)
    _check(dw.cpu().numpy(), dw_ref, "dw", tp, tp2)
    _check(db.cpu().numpy(), db_ref, "db", tp, tp2, floor=_zero_floor(st, gy) if mode == 2 else 1e-6)
    if mode:
        _check(dg.cpu().numpy(), dg_ref, "dgamma", tp, tp2)
        _check(dbt.cpu().numpy(), dbt_ref, "dbeta", tp, tp2)


@pytest.mark.parametrize("st", ["f32", "bf16"])
@pytest.mark.parametrize("cin,cout,n,h,w,mode", [(128, 64, 2, 6, 10, 0), (128, 64, 2, 6, 10, 2), (64, 16, 2, 8, 12, 1),
                                                 (64, 16, 1, 10, 16, 2), (64, 16, 2, 8, 12, 0)])
def test_upsampler_backward_strict(st, cin, cout, n, h, w, mode):
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    dt = BF if st == "bf16" else torch.float32
    tag = f"strict_up_{cin}_{cout}_{mode}"
    x = np.maximum(gu.det_normal(tag + ".x", (n, cin, h, w)), 0)
    wt = gu.det_normal(tag + ".w", (cin, cout, 3, 3)) * (1.0 / (4 * cin) ** 0.5)
    b = 0.1 * gu.det_normal(tag + ".b", (cout,))
    gy = gu.det_normal(tag + ".gy", (n, cout, 2 * h, 2 * w))
    if st == "bf16":
        x, wt, gy = _rb(x), _rb(wt), _rb(gy)
    bn = _Bn(tag, cout, mode)
    y_ref, pre_ref = orc.upsampler_fwd(x, wt, b, bn.orc)
    dx_ref, dw_ref, db_ref, dg_ref, dbt_ref = orc.upsampler_bwd(x, wt, y_ref, pre_ref, gy, bn.orc)

    xd, yd, gyd = _dev_nhwc(x, dt), _dev_nhwc(y_ref, dt), _dev_nhwc(gy, dt)
    pred = None if pre_ref is None else _dev_nhwc(pre_ref, dt)
    wd = torch.from_numpy(wt).to(DEV)
    dx = torch.empty_like(xd)
    dw, db = torch.empty_like(wd), torch.empty(cout, device=DEV)
    dg, dbt = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    wsb = torch.empty(int(lib.erf_workspace_bytes(n * 4 * h * w, max(cin, cout))), dtype=torch.uint8, device=DEV)
    bref = bn.struct(None if not mode else bn.orc.save_mean, None if not mode else bn.orc.save_invstd)
    before = _lib.launch_count()
    _lib.check(_fn(lib, "erf_upsampler_backward", st)(
        _p(xd), _p(wd), bref, _p(yd), _p(pred), _p(gyd), _p(dx), _p(dw), _p(db), _p(dg) if mode else None,
        _p(dbt) if mode else None, n, cin, h, w, cout, 0, _p(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    assert _lib.launch_count() > before
    tx, tx2, tp, tp2 = _tol(st, mode)
    _check(_host_nchw(dx, x.shape), dx_ref, "dx", tx, tx2)
    _check(dw.cpu().numpy(), dw_ref, "dw", tp, tp2)
    _check(db.cpu().numpy(), db_ref, "db", tp, tp2, floor=_zero_floor(st, gy) if mode == 2 else 1e-6)
    if mode:
        _check(dg.cpu().numpy(), dg_ref, "dgamma", tp, tp2)
        _check(dbt.cpu().numpy(), dbt_ref, "dbeta", tp, tp2)
    # the flags of the mask hand-over: dx masked by (x > 0), parameter gradients unchanged
    dx2 = torch.empty_like(xd)
    dw2, db2 = torch.empty_like(wd), torch.empty(cout, device=DEV)
    _lib.check(_fn(lib, "erf_upsampler_backward", st)(
        _p(xd), _p(wd), bref, _p(yd), _p(pred), _p(gyd), _p(dx2), _p(dw2), _p(db2), _p(dg) if mode else None,
        _p(dbt) if mode else None, n, cin, h, w, cout, _lib.BWD_MASK_DX, _p(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert torch.equal(dx2, dx * (xd > 0))


@pytest.mark.parametrize("st", ["f32", "bf16"])
@pytest.mark.parametrize("ncls,n,h,w", [(20, 2, 10, 16), (1, 2, 12, 12), (19, 1, 16, 24)])
def test_outconv_backward_strict(st, ncls, n, h, w):
    """ConvTranspose2d(16, ncls, 2, stride=2): the gradient of the logits (fp32, NCHW) in, dx (storage
    type, NHWC), dW, db out."""
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    dt = BF if st == "bf16" else torch.float32
    tag = f"strict_outconv_{ncls}"
    cin = 16
    x = np.maximum(gu.det_normal(tag + ".x", (n, cin, h, w)), 0)
    wt = gu.det_normal(tag + ".w", (cin, ncls, 2, 2)) * 0.25
    gy = gu.det_normal(tag + ".gy", (n, ncls, 2 * h, 2 * w))
    if st == "bf16":
        x, wt = _rb(x), _rb(wt)
    dx_ref, dw_ref, db_ref = orc.convT2d_bwd(x, wt, gy, 2, 0, 0)
    xd = _dev_nhwc(x, dt)
    wd, gyd = torch.from_numpy(wt).to(DEV), torch.from_numpy(gy).to(DEV)
    dx, dw, db = torch.empty_like(xd), torch.empty_like(wd), torch.empty(ncls, device=DEV)
    wsb = torch.empty(int(lib.erf_workspace_bytes(n * 4 * h * w, 32)), dtype=torch.uint8, device=DEV)
    _lib.check(_fn(lib, "erf_outconv_backward", st)(_p(xd), _p(wd), _p(gyd), _p(dx), _p(dw), _p(db), n, cin, h, w, ncls,
                                                    0, _p(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    tx, tx2, tp, tp2 = TOL[st]
    if st == "bf16":
        tx, tx2 = 2.0 ** -8, 3e-3            # one rounding of dx
    _check(_host_nchw(dx, x.shape), dx_ref, "dx", tx, tx2)
    _check(dw.cpu().numpy(), dw_ref, "dw", tp, tp2)
    _check(db.cpu().numpy(), db_ref, "db", tp, tp2)


@pytest.mark.parametrize("c,dil,n,h,w,mode,drop", [(16, 1, 2, 18, 32, 0, False), (16, 1, 2, 17, 33, 0, False),
                                                   (16, 1, 2, 18, 32, 2, False), (64, 2, 2, 20, 24, 2, False),
                                                   (128, 4, 2, 12, 20, 2, True), (64, 8, 1, 20, 24, 1, True),
                                                   (16, 1, 1, 16, 48, 1, True)])
def test_nb1d_backward_bf16_strict_all_widths_and_batchnorm(c, dil, n, h, w, mode, drop):
    """non_bottleneck_1d (train/erfnet_blocks.py:33-73) on bf16 storage: 16 channels (both 16-channel
    kernels: image width a multiple of 16 or not), BatchNorm eval / train, Dropout2d factors."""
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    tag = f"strict_nb1d_bf16_{c}_{dil}_{mode}_{w}"
    x = _rb(np.maximum(gu.det_normal(tag + ".x", (n, c, h, w)), 0))
    shapes = [(c, c, 3, 1), (c, c, 1, 3), (c, c, 3, 1), (c, c, 1, 3)]
    ws = [_rb(gu.det_normal(f"{tag}.w{i}", s) * (1.0 / (3 * c) ** 0.5)) for i, s in enumerate(shapes)]
    bs = [0.1 * gu.det_normal(f"{tag}.b{i}", (c,)) for i in range(4)]
    bns = [_Bn(f"{tag}.bn{i}", c, mode) for i in (1, 2)]
    ds = None
    if drop:
        ds = ((gu.det_uniform(tag + ".keep", (n, c), 0, 1) >= 0.3) / 0.7).astype(np.float32)
    # the forward the kernels would have stored: every saved activation rounded to bf16 before its consumer
    # reads it is NOT what the oracle does (it keeps fp32), so the point of differentiation differs by one
    # rounding per tensor; the bounds above leave room for exactly that
    y_ref, sv = orc.nb1d_fwd(x, ws, bs, dil, bns[0].orc, bns[1].orc, ds, save=True)
    gy = _rb(gu.det_normal(tag + ".gy", y_ref.shape))
    dx_ref, dws_ref, dbs_ref, dgs_ref, dbts_ref = orc.nb1d_bwd(x, ws, dil, y_ref, sv, gy, bns[0].orc, bns[1].orc, ds)

    xd, yd, gyd = _dev_nhwc(x, BF), _dev_nhwc(y_ref, BF), _dev_nhwc(gy, BF)
    save = [_dev_nhwc(sv[i], BF) for i in range(5)]
    wd = [torch.from_numpy(np.ascontiguousarray(w_.reshape(c, c, 3))).to(DEV) for w_ in ws]
    dsd = None if ds is None else torch.from_numpy(ds).to(DEV)
    dx = torch.empty_like(xd)
    dw = [torch.empty(c, c, 3, device=DEV) for _ in range(4)]
    db = [torch.empty(c, device=DEV) for _ in range(4)]
    dg = [torch.empty(c, device=DEV) for _ in range(2)]
    dbt = [torch.empty(c, device=DEV) for _ in range(2)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[None if t_ is None else t_.data_ptr() for t_ in ts])
    wsb = torch.empty(int(lib.erf_workspace_bytes(n * h * w, c)), dtype=torch.uint8, device=DEV)
    brefs = [b_.struct(None if not mode else b_.orc.save_mean, None if not mode else b_.orc.save_invstd) for b_ in bns]
    _lib.check(lib.erf_nb1d_backward_bf16(_p(xd), arr(wd), brefs[0], brefs[1], _p(dsd), dil, _p(yd), arr(save), _p(gyd),
                                          _p(dx), arr(dw), arr(db), arr(dg) if mode else None,
                                          arr(dbt) if mode else None, n, c, h, w, 0, _p(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    tx, tx2 = 2.0 ** -6, 1e-2                                     # four chained roundings of the gradient
    tp, tp2 = (1e-2, 6e-3) if mode else (6e-3, 4e-3)
    _check(_host_nchw(dx, x.shape), dx_ref, "dx", tx, tx2)
    for i in range(4):
        _check(dw[i].cpu().numpy().reshape(ws[i].shape), dws_ref[i], f"dw{i}", tp, tp2)
        _check(db[i].cpu().numpy(), dbs_ref[i], f"db{i}", tp, tp2,
               floor=_zero_floor("bf16", gy) if (mode == 2 and i in (1, 3)) else 1e-6)
    if mode:
        for i in range(2):
            _check(dg[i].cpu().numpy(), dgs_ref[i], f"dgamma{i}", tp, tp2)
            _check(dbt[i].cpu().numpy(), dbts_ref[i], f"dbeta{i}", tp, tp2)
