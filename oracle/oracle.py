"""numpy/ctypes front-end of the plain-C CPU oracle (``erfnet_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / timed CPU baseline.  The product package
(``erfnet_pytorch_amd``) never imports it.

All tensors are NCHW float32 numpy arrays, as in the reference
(``train/erfnet_blocks.py``, ``eval/erfnet.py``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liberfnet_oracle.so")
_lib = None

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


class _BN(C.Structure):
    _fields_ = [("mode", C.c_int), ("gamma", _fp), ("beta", _fp), ("run_mean", _fp),
                ("run_var", _fp), ("save_mean", _fp), ("save_invstd", _fp),
                ("momentum", C.c_float), ("eps", C.c_float)]


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "erfnet_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liberfnet_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_downsampler_fwd.restype = C.c_int
        _lib.orc_downsampler_bwd.restype = C.c_int
        _lib.orc_erfnet_forward.restype = C.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


class BN:
    """BatchNorm2d parameter pack (``nn.BatchNorm2d(C, eps=1e-3)``, train/erfnet_blocks.py:23).

    mode: 0 off, 1 eval (running stats), 2 train (batch stats; running stats updated in place).
    """

    def __init__(self, mode, gamma=None, beta=None, running_mean=None, running_var=None,
                 momentum=0.1, eps=1e-3):
        self.mode = int(mode)
        self.gamma = None if gamma is None else _f(gamma).copy()
        self.beta = None if beta is None else _f(beta).copy()
        self.running_mean = None if running_mean is None else _f(running_mean).copy()
        self.running_var = None if running_var is None else _f(running_var).copy()
        c = 0 if self.gamma is None else self.gamma.size
        self.save_mean = np.zeros(max(c, 1), np.float32)
        self.save_invstd = np.zeros(max(c, 1), np.float32)
        self.momentum, self.eps = float(momentum), float(eps)

    def struct(self):
        return _BN(self.mode, _p(self.gamma), _p(self.beta), _p(self.running_mean),
                   _p(self.running_var), _p(self.save_mean), _p(self.save_invstd),
                   self.momentum, self.eps)


_NOBN = None


def _bnref(bn):
    if bn is None or bn.mode == 0:
        return None
    s = bn.struct()
    bn._keep = s
    return C.byref(s)


# ------------------------------------------------------------------------------ primitives
def conv2d_fwd(x, w, b, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    Ho = (H + 2 * padding[0] - dilation[0] * (KH - 1) - 1) // stride[0] + 1
    Wo = (W + 2 * padding[1] - dilation[1] * (KW - 1) - 1) // stride[1] + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, H, W, Cout, KH, KW, stride[0],
                         stride[1], padding[0], padding[1], dilation[0], dilation[1])
    return y


def conv2d_bwd(x, w, gy, stride=(1, 1), padding=(0, 0), dilation=(1, 1), need_dx=True):
    x, w, gy = _f(x), _f(w), _f(gy)
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty(Cout, np.float32)
    lib().orc_conv2d_bwd(_p(x), _p(w), _p(gy), _p(dx), _p(dw), _p(db), N, Cin, H, W, Cout, KH,
                         KW, stride[0], stride[1], padding[0], padding[1], dilation[0],
                         dilation[1])
    return dx, dw, db


def convT2d_fwd(x, w, b, stride, padding, output_padding):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    N, Cin, H, W = x.shape
    _, Cout, KH, KW = w.shape
    Ho = (H - 1) * stride - 2 * padding + KH + output_padding
    Wo = (W - 1) * stride - 2 * padding + KW + output_padding
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_convT2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, H, W, Cout, KH, KW, stride,
                          padding, output_padding)
    return y


def convT2d_bwd(x, w, gy, stride, padding, output_padding, need_dx=True):
    x, w, gy = _f(x), _f(w), _f(gy)
    N, Cin, H, W = x.shape
    _, Cout, KH, KW = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty(Cout, np.float32)
    lib().orc_convT2d_bwd(_p(x), _p(w), _p(gy), _p(dx), _p(dw), _p(db), N, Cin, H, W, Cout, KH,
                          KW, stride, padding, output_padding)
    return dx, dw, db


def softmax_c(x):
    x = _f(x)
    N, Cc, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_softmax_c(_p(x), _p(y), N, Cc, H * W)
    return y


# ---------------------------------------------------------------------------------- blocks
def downsampler_fwd(x, w, b, bn=None):
    """DownsamplerBlock.forward (train/erfnet_blocks.py:25-29).  Returns (y, pre_bn)."""
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, H, W = x.shape
    Cout = w.shape[0] + Cin
    y = np.empty((N, Cout, H // 2, W // 2), np.float32)
    pre = np.empty_like(y) if (bn is not None and bn.mode) else None
    rc = lib().orc_downsampler_fwd(_p(x), _p(w), _p(b), _bnref(bn), _p(y), _p(pre), N, Cin, H,
                                   W, Cout)
    if rc != 0:
        raise RuntimeError("DownsamplerBlock needs even H and W (torch.cat size mismatch in "
                           "the reference)")
    return y, pre


def downsampler_bwd(x, w, y, pre, gy, bn=None, need_dx=True):
    x, w, y, gy = _f(x), _f(w), _f(y), _f(gy)
    N, Cin, H, W = x.shape
    Cout = w.shape[0] + Cin
    dx = np.empty_like(x) if need_dx else None
    dw, db = np.empty_like(w), np.empty(w.shape[0], np.float32)
    has_bn = bn is not None and bn.mode
    dg = np.empty(Cout, np.float32) if has_bn else None
    dbt = np.empty(Cout, np.float32) if has_bn else None
    lib().orc_downsampler_bwd(_p(x), _p(w), _bnref(bn), _p(y), _p(pre), _p(gy), _p(dx), _p(dw),
                              _p(db), _p(dg), _p(dbt), N, Cin, H, W, Cout)
    return dx, dw, db, dg, dbt


def _arr4(lst):
    arr = (_fp * len(lst))(*[_p(a) for a in lst])
    return arr


def nb1d_fwd(x, ws, bs, dil, bn1=None, bn2=None, drop_scale=None, save=False):
    """non_bottleneck_1d.forward (train/erfnet_blocks.py:55-73).

    ws: [conv3x1_1.weight, conv1x3_1.weight, conv3x1_2.weight, conv1x3_2.weight]
    Returns y or (y, saved) where saved = [t1, pre2, t2, t3, pre4, t4bn].
    """
    x = _f(x)
    ws = [_f(w).reshape(w.shape[0], w.shape[1], 3) for w in ws]
    bs = [_f(b) for b in bs]
    N, Cc, H, W = x.shape
    y = np.empty_like(x)
    sv = [np.empty_like(x) for _ in range(6)] if save else None
    ds = None if drop_scale is None else _f(drop_scale)
    lib().orc_nb1d_fwd(_p(x), _arr4(ws), _arr4(bs), _bnref(bn1), _bnref(bn2), _p(ds), int(dil),
                       _p(y), None if sv is None else _arr4(sv), N, Cc, H, W)
    return (y, sv) if save else y


def nb1d_bwd(x, ws, dil, y, saved, gy, bn1=None, bn2=None, drop_scale=None):
    x, y, gy = _f(x), _f(y), _f(gy)
    shapes = [w.shape for w in ws]
    ws = [_f(w).reshape(w.shape[0], w.shape[1], 3) for w in ws]
    N, Cc, H, W = x.shape
    dx = np.empty_like(x)
    dws = [np.empty_like(w) for w in ws]
    dbs = [np.empty(Cc, np.float32) for _ in range(4)]
    dgs = [np.zeros(Cc, np.float32) for _ in range(2)]
    dbts = [np.zeros(Cc, np.float32) for _ in range(2)]
    ds = None if drop_scale is None else _f(drop_scale)
    lib().orc_nb1d_bwd(_p(x), _arr4(ws), _bnref(bn1), _bnref(bn2), _p(ds), int(dil), _p(y),
                       _arr4(saved), _p(gy), _p(dx), _arr4(dws), _arr4(dbs), _arr4(dgs),
                       _arr4(dbts), N, Cc, H, W)
    dws = [d.reshape(s) for d, s in zip(dws, shapes)]
    return dx, dws, dbs, dgs, dbts


def upsampler_fwd(x, w, b, bn=None):
    """UpsamplerBlock.forward (train/erfnet_blocks.py:84-88).  Returns (y, pre_bn)."""
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, H, W = x.shape
    Cout = w.shape[1]
    y = np.empty((N, Cout, 2 * H, 2 * W), np.float32)
    pre = np.empty_like(y) if (bn is not None and bn.mode) else None
    lib().orc_upsampler_fwd(_p(x), _p(w), _p(b), _bnref(bn), _p(y), _p(pre), N, Cin, H, W, Cout)
    return y, pre


def upsampler_bwd(x, w, y, pre, gy, bn=None):
    x, w, y, gy = _f(x), _f(w), _f(y), _f(gy)
    N, Cin, H, W = x.shape
    Cout = w.shape[1]
    dx = np.empty_like(x)
    dw, db = np.empty_like(w), np.empty(Cout, np.float32)
    has_bn = bn is not None and bn.mode
    dg = np.empty(Cout, np.float32) if has_bn else None
    dbt = np.empty(Cout, np.float32) if has_bn else None
    lib().orc_upsampler_bwd(_p(x), _p(w), _bnref(bn), _p(y), _p(pre), _p(gy), _p(dx), _p(dw),
                            _p(db), _p(dg), _p(dbt), N, Cin, H, W, Cout)
    return dx, dw, db, dg, dbt


# ------------------------------------------------------------------------------- whole net
def erfnet_param_order(prefix_style):
    """Names of (weight, bias) tensors in the order ``orc_erfnet_forward`` consumes them.

    prefix_style 'eval'  : eval/erfnet.py, eval/erfnet_nobn.py state_dict keys
                 'train' : train/net_classification.py / net_regression.py keys
    """
    names, bns = [], []

    def nb(p):
        for c in ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2"):
            names.extend([f"{p}.{c}.weight", f"{p}.{c}.bias"])
        bns.extend([f"{p}.bn1", f"{p}.bn2"])

    def cv(p):
        names.extend([f"{p}.conv.weight", f"{p}.conv.bias"])
        bns.append(f"{p}.bn")

    cv("encoder.initial_block")
    cv("encoder.layers.0")
    for i in range(1, 6):
        nb(f"encoder.layers.{i}")
    cv("encoder.layers.6")
    for i in range(7, 15):
        nb(f"encoder.layers.{i}")
    if prefix_style == "eval":
        cv("decoder.layers.0"); nb("decoder.layers.1"); nb("decoder.layers.2")
        cv("decoder.layers.3"); nb("decoder.layers.4"); nb("decoder.layers.5")
        names.extend(["decoder.output_conv.weight", "decoder.output_conv.bias"])
    else:
        for d in ("scalar_decoder_1", "scalar_decoder_2"):
            cv(f"decoder.{d}.layers.0"); nb(f"decoder.{d}.layers.1"); nb(f"decoder.{d}.layers.2")
        names.extend(["decoder.scalar_output_conv.weight", "decoder.scalar_output_conv.bias"])
    return names, bns


def erfnet_forward(x, state, prefix_style="eval", use_bn=False, ncls=None, bn_eps=1e-3):
    """Whole-net forward (eval mode).  ``state``: dict key -> numpy array (a state_dict)."""
    x = _f(x)
    names, bns = erfnet_param_order(prefix_style)
    arrs = [_f(state[k]) for k in names]
    if ncls is None:
        ncls = arrs[-1].shape[0]
    params = (_fp * len(arrs))(*[_p(a) for a in arrs])
    bnp = None
    keep = []
    if use_bn:
        for b in bns:
            for s in ("weight", "bias", "running_mean", "running_var"):
                keep.append(_f(state[f"{b}.{s}"]))
        bnp = (_fp * len(keep))(*[_p(a) for a in keep])
    N, _, H, W = x.shape
    out = np.empty((N, ncls, H, W), np.float32)
    rc = lib().orc_erfnet_forward(_p(x), params, bnp, C.c_float(bn_eps), _p(out), N, H, W, ncls)
    if rc != 0:
        raise RuntimeError("H and W must be multiples of 8")
    return out


# ------------------------------------------------------------- whole-network training pass (no BN)
def _ce_loss_and_grad(logits, labels, ignore_index=-1):
    """nn.CrossEntropyLoss(ignore_index=-1) (train/evaluation_functions.py:6-14): mean over the
    labelled pixels of -log softmax; returns (loss, dloss/dlogits)."""
    n, c, h, w = logits.shape
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    valid = labels != ignore_index
    cnt = max(int(valid.sum()), 1)
    lab = np.where(valid, labels, 0)
    picked = np.take_along_axis(logp, lab[:, None], axis=1)[:, 0]
    loss = -(picked * valid).sum() / cnt
    g = np.exp(logp)
    np.put_along_axis(g, lab[:, None], np.take_along_axis(g, lab[:, None], axis=1) - 1.0, axis=1)
    g *= (valid[:, None] / cnt)
    return float(loss), g.astype(np.float32)


def erfnet_train_pass(x, labels, state, prefix_style="train"):
    """Forward + CrossEntropy + backward of the BN-free ERFNet (train/net_classification.py:41-66,
    104-110 driven as in train/main.py:316-379), composed from the block functions above.
    Returns (loss, {state_dict key: gradient})."""
    x = _f(x)
    names, _ = erfnet_param_order(prefix_style)
    P = [_f(state[k]) for k in names]
    plan = [("down",), ("down",)] + [("nb", 1)] * 5 + [("down",)] + \
        [("nb", d) for _ in range(2) for d in (2, 4, 8, 16)] + \
        [("up",), ("nb", 1), ("nb", 1), ("up",), ("nb", 1), ("nb", 1)]
    tape, cur, pi = [], x, 0
    for op in plan:
        if op[0] == "nb":
            ws, bs = P[pi:pi + 8:2], P[pi + 1:pi + 8:2]
            y, sv = nb1d_fwd(cur, ws, bs, op[1], save=True)
            tape.append(("nb", cur, ws, op[1], y, sv, pi))
            pi += 8
        elif op[0] == "down":
            y, _ = downsampler_fwd(cur, P[pi], P[pi + 1])
            tape.append(("down", cur, P[pi], y, pi))
            pi += 2
        else:
            y, _ = upsampler_fwd(cur, P[pi], P[pi + 1])
            tape.append(("up", cur, P[pi], y, pi))
            pi += 2
        cur = y
    logits = convT2d_fwd(cur, P[pi], P[pi + 1], 2, 0, 0)
    loss, g = _ce_loss_and_grad(logits, labels)
    grads = {}
    g, dw, db = convT2d_bwd(cur, P[pi], g, 2, 0, 0)
    grads[names[pi]], grads[names[pi + 1]] = dw, db
    for rec in reversed(tape):
        if rec[0] == "nb":
            _, xin, ws, d, y, sv, i = rec
            g, dws, dbs, _, _ = nb1d_bwd(xin, ws, d, y, sv, g)
            for j in range(4):
                grads[names[i + 2 * j]] = dws[j].reshape(state[names[i + 2 * j]].shape)
                grads[names[i + 2 * j + 1]] = dbs[j]
        elif rec[0] == "down":
            _, xin, w, y, i = rec
            g, dw, db, _, _ = downsampler_bwd(xin, w, y, None, g, need_dx=i != 0)
            grads[names[i]], grads[names[i + 1]] = dw, db
        else:
            _, xin, w, y, i = rec
            g, dw, db, _, _ = upsampler_bwd(xin, w, y, None, g)
            grads[names[i]], grads[names[i + 1]] = dw, db
    return loss, grads


# ---------------------------------------------------------------------------- masked regression losses
def masked_loss(kind, a, b, c, target, eps=0.0):
    """float64 numpy restatement of the reference's masked regression losses and their gradients
    (train/evaluation_functions.py): kind "l1" :89-97, "ll" :71-87, "l1cp" :32-47, "llcp" :49-69.
    a: [N,K,H,W] (K = 1 for l1 / ll); b: [N,H,W] variance (ll) or [K] cost; c: [K] variance (llcp);
    target: [N,H,W], -1 = no label.  Returns (loss, da, db, dc)."""
    a = np.asarray(a, np.float64)
    t = np.asarray(target, np.float64)[:, None]                     # [N,1,H,W]
    valid = t != -1.0
    cnt = valid.sum()
    g = valid / cnt if cnt else valid * np.nan
    db = dc = None
    if kind == "l1":
        v = np.abs(a - t)
        da = np.sign(a - t) * g
    elif kind == "ll":
        s = np.asarray(b, np.float64)[:, None] + eps
        e = a - t
        v = np.log(2 * np.pi * s) + e * e / s
        da = 2 * e / s * g
        db = ((1 / s - e * e / (s * s)) * g)[:, 0]
    else:
        cost = np.asarray(b, np.float64).reshape(1, -1, 1, 1)
        e = cost - t
        if kind == "l1cp":
            wk = np.abs(e)
            db = (a * np.sign(e) * g).sum(axis=(0, 2, 3))
        else:
            s = np.asarray(c, np.float64).reshape(1, -1, 1, 1) + eps
            wk = np.log(2 * np.pi * s) + e * e / s
            db = (a * 2 * e / s * g).sum(axis=(0, 2, 3))
            dc = (a * (1 / s - e * e / (s * s)) * g).sum(axis=(0, 2, 3))
        v = (wk * a).sum(axis=1, keepdims=True)
        da = wk * g
    loss = v[valid].mean() if cnt else np.nan
    return loss, da, db, dc


def mse_weighted(a, b, scale_factor, weight):
    """MSELossWeighted (train/evaluation_functions.py:122-131): value and gradient w.r.t. a."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    e = a - b
    return (e * e * scale_factor * weight).mean(), 2 * e * scale_factor * weight / e.size
