"""``torch.autograd.Function`` wrappers over the C ABI of ``liberfnet_hip.so``.

PyTorch is plumbing here: it owns device memory, streams and the autograd graph; every number a
GPU tensor receives is computed by the HIP kernels.  Activations are kept NHWC in memory (logical
shape stays [N,C,H,W], i.e. ``torch.channels_last``); the classifier logits leave the output
transposed convolution NCHW-contiguous so that torch's loss / argmax read them directly.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import BnStruct, vp

_workspaces = {}


def _dev_ctx(t):
    return torch.cuda.device(t.device)


def _stream(t):
    return vp(torch.cuda.current_stream(t.device).cuda_stream)


def _workspace(t, n_pixels, max_c, copies=1):
    """Scratch buffer for one call, cached per (device, stream); stream order keeps reuse safe.
    copies > 1: room for that many disjoint workspaces (:func:`_workspace_slice` bytes apart)."""
    need = _workspace_slice(n_pixels, max_c) * int(copies)
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < need:
        buf = None
        _workspaces.pop(key, None)
        buf = torch.empty(int(need * 1.25), dtype=torch.uint8, device=t.device)
        _workspaces[key] = buf
    return buf


def _workspace_slice(n_pixels, max_c):
    """Bytes of one call's workspace, rounded up to a multiple of 256."""
    return (_lib.lib().erf_workspace_bytes(int(n_pixels), int(max_c)) + 255) // 256 * 256


def _p(t):
    return None if t is None else vp(t.data_ptr())


def _is_nhwc_dense(x):
    return x.permute(0, 2, 3, 1).is_contiguous()


_ACT_DTYPES = (torch.float32, torch.bfloat16)


def _check_act(x):
    if x.dtype not in _ACT_DTYPES:
        raise TypeError(f"erfnet_hip kernels store activations as fp32 or bf16, got {x.dtype}")


def _nhwc(x, dtype=None):
    """Dense NHWC view/copy of a logical-NCHW tensor (fp32 or bf16; ``dtype`` forces the storage
    type, e.g. a gradient arriving in fp32 for bf16 activations)."""
    _check_act(x)
    if dtype is not None and x.dtype != dtype:
        out = torch.empty(x.shape, dtype=dtype, device=x.device, memory_format=torch.channels_last)
        out.copy_(x)
        return out
    if _is_nhwc_dense(x):
        return x
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    out.copy_(x)
    return out


def _empty_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), dtype=like.dtype, device=like.device,
                       memory_format=torch.channels_last)


def _fn(name, x):
    """Entry point for the storage type of ``x``: ``name`` for fp32, ``name_bf16`` for bf16."""
    return getattr(_lib.lib(), name + "_bf16" if x.dtype == torch.bfloat16 else name)


# Parameters are handed to the library in fp32 (the kernels repack / round them per call).  A model
# converted with .bfloat16() is served through fp32 copies, refreshed when a parameter's version
# counter moves (optimiser steps, copy_, load_state_dict) or its storage is replaced (the reference's
# `target.data = ...`, train/main.py:142).  An in-place update through a `.data` alias
# (`p.data.mul_(a)`) moves neither: code that does that calls invalidate_parameter_copies().
_fp32_copies = {}


def invalidate_parameter_copies():
    """Drop the fp32 copies kept of non-fp32 parameters (see above); the next call converts afresh."""
    _fp32_copies.clear()


def _as_fp32(t):
    key = (t.device, t.data_ptr(), tuple(t.shape))
    hit = _fp32_copies.get(key)
    if hit is not None and hit[0] == t._version and hit[1].dtype == t.dtype:
        return hit[2]
    if len(_fp32_copies) > 4096:
        _fp32_copies.clear()
    c = t.detach().to(torch.float32).contiguous()
    _fp32_copies[key] = (t._version, t, c)
    return c


def _param(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        return _as_fp32(t)
    return t.detach().contiguous()


def _plain(t):
    """A parameter whose memory is handed to the library as it is (fp32, dense)."""
    if t.dtype != torch.float32:
        return _as_fp32(t)
    return t if t.is_contiguous() else t.detach().contiguous()


class _Bn:
    """Per-call BatchNorm state handed to the library (mode 0 = off, 1 = eval, 2 = train)."""

    def __init__(self, mode, gamma, beta, running_mean, running_var, momentum, eps, need_save,
                 saved=None):
        self.mode = int(mode)
        self.gamma, self.beta = _param(gamma), _param(beta)
        if running_mean is not None and running_mean.dtype != torch.float32:
            if self.mode == 2:
                raise TypeError("BatchNorm in train mode needs fp32 running statistics (keep the module in fp32 "
                                "and feed bf16 activations)")
            running_mean, running_var = _as_fp32(running_mean), _as_fp32(running_var)
        self.rm, self.rv = running_mean, running_var
        self.momentum, self.eps = float(momentum), float(eps)
        self.save_mean = self.save_invstd = None
        if saved is not None:
            self.save_mean, self.save_invstd = saved
        elif self.mode == 2 or (self.mode and need_save):     # batch statistics always land here
            c = gamma.numel()
            self.save_mean = torch.empty(c, dtype=torch.float32, device=gamma.device)
            self.save_invstd = torch.empty(c, dtype=torch.float32, device=gamma.device)
        # nn.BatchNorm2d's step counter rides on the running_mean tensor (erfnet_blocks._bn_args);
        # the statistics kernel of a train-mode forward increments it
        self.nbt = getattr(running_mean, "_erf_nbt", None) if self.mode == 2 else None
        self.struct = BnStruct(self.mode, _p(self.gamma), _p(self.beta), _p(self.rm), _p(self.rv),
                               _p(self.save_mean), _p(self.save_invstd), self.momentum, self.eps, _p(self.nbt))

    def ref(self):
        return C.byref(self.struct) if self.mode else None


class _NoBn:
    """Stand-in for an absent BatchNorm (mode 0)."""
    mode = 0
    gamma = beta = save_mean = save_invstd = None

    @staticmethod
    def ref():
        return None


_NO_BN = _NoBn()


def _arr(ts):
    return (vp * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class _InferenceCtx:
    """Minimal stand-in for an autograd context: under torch.no_grad() the block functions are called
    directly, which skips the autograd machinery of Function.apply (tens of microseconds per call,
    a third of the host time of a batch-1 forward)."""

    def __init__(self, n_inputs):
        self.needs_input_grad = (False,) * n_inputs

    def save_for_backward(self, *tensors):
        pass


def _apply(fn, *args):
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_InferenceCtx(len(args)), *args)


# ReLU-mask hand-over between neighbouring blocks.  Every block output is a ReLU output, and the
# first thing a block's backward does is mask the incoming gradient with (output > 0).  When the
# input of block k is the output of block k-1, block k applies that mask to the dx it returns
# (fused into its data-gradient epilogue) and leaves the tensor in the token both blocks share;
# block k-1 skips its masking pass when the gradient autograd hands it IS that tensor, i.e. when
# no other consumer contributed to it.  Masking is idempotent, so every other case (no token,
# several consumers, a copy in between) simply takes the ordinary path.
class _ReluToken:
    __slots__ = ("dx", "grad_on")

    def __init__(self):
        self.dx = None
        # Function.forward always runs with grad mode off and ctx.needs_input_grad ignores
        # torch.no_grad(), so the wrapper (which creates the token) records the caller's mode:
        # inference under no_grad keeps no activations and folds eval-mode BatchNorm
        self.grad_on = torch.is_grad_enabled()


def _input_token(x):
    return getattr(x, "_erf_relu_token", None)


def _tag_output(y, token):
    y._erf_relu_token = token
    return y


def _bwd_flags(ctx, gy):
    """Flags for a backward call: is gy the pre-masked dx of the single consumer; may dx be masked."""
    flags = 0
    tok = ctx.out_token
    if tok is not None and tok.dx is not None:
        d = tok.dx
        if d.data_ptr() == gy.data_ptr() and d.shape == gy.shape and d.stride() == gy.stride():
            flags |= _lib.BWD_GY_PREMASKED
        tok.dx = None
    if ctx.in_token is not None:
        flags |= _lib.BWD_MASK_DX
    return flags


def _hand_over(ctx, dx):
    if ctx.in_token is not None and dx is not None:
        ctx.in_token.dx = dx
    return dx


# Direct gradient write.  An owner of the parameters (ddp.GradientReducer) may attach
# ``_erf_grad_sink = (buffer, owner)`` to a Parameter: backward then lets the weight-gradient kernel
# write straight into ``buffer`` (the parameter's slice of a flat bucket) and returns None to
# autograd, which saves the temporary and autograd's accumulation kernel.  The owner grants the
# buffer once per step (``_claim``); any further gradient of the same parameter in that step takes
# the ordinary accumulate path.  ``_delivered`` tells the owner the write has been enqueued.
def _grad_targets(params, shapes, device):
    """[(tensor the kernel writes, parameter if it is the owner's buffer else None)]; a None shape
    gives (None, None)."""
    out = []
    for q, shape in zip(params, shapes):
        if shape is None:
            out.append((None, None))
            continue
        sink = getattr(q, "_erf_grad_sink", None) if q is not None else None
        if sink is not None and tuple(sink[0].shape) == tuple(shape) and sink[1]._claim(q):
            out.append((sink[0], q))
        else:
            out.append((torch.empty(tuple(shape), dtype=torch.float32, device=device), None))
    return out


def _deliver(targets):
    """Values handed to autograd: None where the kernel wrote into the owner's buffer."""
    res = []
    for buf, q in targets:
        if q is not None:
            q._erf_grad_sink[1]._delivered(q)
            res.append(None)
        else:
            res.append(buf)
    return res


def _match_params(grads, params):
    """Gradients come out of the kernels in fp32; a parameter kept in another dtype gets a cast."""
    return tuple(g if (g is None or p is None or g.dtype == p.dtype) else g.to(p.dtype)
                 for g, p in zip(grads, params))


# ------------------------------------------------------------------------------ DownsamplerBlock
class _Downsampler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rm, rv, mode, momentum, eps, out_token):
        lib = _lib.lib()
        n, cin, h, wd = x.shape
        cout = w.shape[0] + cin
        ctx.in_token = _input_token(x) if cin != 3 else None
        ctx.out_token = out_token
        if cin != 3 or ctx.needs_input_grad[0]:
            x = _nhwc(x)     # a 3-channel image is read through its strides unless dx is wanted
        else:
            _check_act(x)
        need = out_token.grad_on and any(ctx.needs_input_grad)
        bn = _Bn(mode, gamma, beta, rm, rv, momentum, eps, need)
        wq, bq = _param(w), _param(b)
        if h % 2 or wd % 2:
            raise RuntimeError(f"DownsamplerBlock needs even H and W, got {h}x{wd} (the reference "
                               "fails in torch.cat)")
        y = _empty_nhwc(n, cout, h // 2, wd // 2, x)
        pre = torch.empty_like(y) if (mode and need) else None
        ws = _workspace(x, n * (h // 2) * (wd // 2), cout)
        sn, sc, sh, sw = x.stride()
        with _dev_ctx(x):
            _lib.check(_fn("erf_downsampler_forward", x)(_p(x), sn, sc, sh, sw, _p(wq), _p(bq), bn.ref(),
                                                   _p(y), _p(pre), n, cin, h, wd, cout, _p(ws),
                                                   ws.numel(), _stream(x)))
        if need:
            ctx.save_for_backward(x, wq, y, pre, bn.gamma, bn.save_mean, bn.save_invstd)
            ctx.cfg = (mode, momentum, eps)
            ctx.params = (w, b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        x, wq, y, pre, gamma, smean, sinv = ctx.saved_tensors
        mode, momentum, eps = ctx.cfg
        n, cin, h, wd = x.shape
        cout = y.shape[1]
        gy = _nhwc(gy, y.dtype)
        bn = _Bn(mode, gamma, None, None, None, momentum, eps, False, saved=(smean, sinv))
        need_dx = ctx.needs_input_grad[0]
        dx = _empty_nhwc(n, cin, h, wd, x) if need_dx else None
        cvec = (cout,) if mode else None
        tg = _grad_targets(ctx.params, (wq.shape, (wq.shape[0],), cvec, cvec), x.device)
        dw, db, dg, dbt = (t[0] for t in tg)
        ws = _workspace(x, n * (h // 2) * (wd // 2), cout)
        sn, sc, sh, sw = x.stride()
        flags = _bwd_flags(ctx, gy) if need_dx else (_bwd_flags(ctx, gy) & _lib.BWD_GY_PREMASKED)
        with _dev_ctx(x):
            _lib.check(_fn("erf_downsampler_backward", x)(_p(x), sn, sc, sh, sw, _p(wq), bn.ref(), _p(y),
                                                    _p(pre), _p(gy), _p(dx), _p(dw), _p(db), _p(dg),
                                                    _p(dbt), n, cin, h, wd, cout, flags, _p(ws),
                                                    ws.numel(), _stream(x)))
        dw, db, dg, dbt = _match_params(_deliver(tg), ctx.params)
        return _hand_over(ctx, dx), dw, db, dg, dbt, None, None, None, None, None, None


def downsampler(x, w, b, bn_weight=None, bn_bias=None, running_mean=None, running_var=None,
                bn_mode=0, momentum=0.1, eps=1e-3):
    """DownsamplerBlock.forward of the reference (train/erfnet_blocks.py:25-29) on the GPU."""
    tok = _ReluToken()
    return _tag_output(_apply(_Downsampler, x, w, b, bn_weight if bn_mode else None,
                                          bn_bias if bn_mode else None, running_mean, running_var,
                                          bn_mode, momentum, eps, tok), tok)


# ------------------------------------------------------------------------------ non_bottleneck_1d
class _Nb1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2, w3, b3, g1, be1, g2, be2, rm1, rv1, rm2, rv2,
                drop_scale, dil, mode, momentum, eps, out_token):
        lib = _lib.lib()
        ctx.in_token = _input_token(x)
        ctx.out_token = out_token
        x = _nhwc(x)
        n, c, h, wd = x.shape
        need = out_token.grad_on and any(ctx.needs_input_grad)
        ws_ = [_param(t) for t in (w0, w1, w2, w3)]
        bs_ = [_param(t) for t in (b0, b1, b2, b3)]
        bn1 = _Bn(mode, g1, be1, rm1, rv1, momentum, eps, need)
        bn2 = _Bn(mode, g2, be2, rm2, rv2, momentum, eps, need)
        ds = None if drop_scale is None else drop_scale.to(torch.float32).contiguous()
        y = torch.empty_like(x)
        save = None
        if need:
            t1, t2, t3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            p2 = torch.empty_like(x) if mode else None
            p4 = torch.empty_like(x) if mode else None
            save = [t1, p2, t2, t3, p4]
        ws = _workspace(x, n * h * wd, c)
        with _dev_ctx(x):
            _lib.check(_fn("erf_nb1d_forward", x)(_p(x), _arr(ws_), _arr(bs_), bn1.ref(), bn2.ref(), _p(ds),
                                            int(dil), _p(y), None if save is None else _arr(save),
                                            n, c, h, wd, _p(ws), ws.numel(), _stream(x)))
        if need:
            ctx.save_for_backward(x, y, *save, *ws_, bn1.gamma, bn2.gamma, bn1.save_mean,
                                  bn1.save_invstd, bn2.save_mean, bn2.save_invstd, ds)
            ctx.cfg = (int(dil), mode, momentum, eps)
            ctx.params = (w0, b0, w1, b1, w2, b2, w3, b3, g1, be1, g2, be2)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        (x, y, t1, p2, t2, t3, p4, w0, w1, w2, w3, g1, g2, sm1, si1, sm2, si2, ds) = ctx.saved_tensors
        dil, mode, momentum, eps = ctx.cfg
        n, c, h, wd = x.shape
        gy = _nhwc(gy, x.dtype)
        bn1 = _Bn(mode, g1, None, None, None, momentum, eps, False, saved=(sm1, si1))
        bn2 = _Bn(mode, g2, None, None, None, momentum, eps, False, saved=(sm2, si2))
        dx = torch.empty_like(x)
        cvec = (c,) if mode else None
        tg = _grad_targets(ctx.params, (w0.shape, (c,), w1.shape, (c,), w2.shape, (c,), w3.shape, (c,),
                                        cvec, cvec, cvec, cvec), x.device)
        dws = [tg[i][0] for i in (0, 2, 4, 6)]
        dbs = [tg[i][0] for i in (1, 3, 5, 7)]
        dgs, dbts = [tg[8][0], tg[10][0]], [tg[9][0], tg[11][0]]
        ws = _workspace(x, n * h * wd, c)
        flags = _bwd_flags(ctx, gy)
        with _dev_ctx(x):
            _lib.check(_fn("erf_nb1d_backward", x)(_p(x), _arr([w0, w1, w2, w3]), bn1.ref(), bn2.ref(), _p(ds),
                                             dil, _p(y), _arr([t1, p2, t2, t3, p4]), _p(gy), _p(dx),
                                             _arr(dws), _arr(dbs), _arr(dgs), _arr(dbts), n, c, h, wd,
                                             flags, _p(ws), ws.numel(), _stream(x)))
        return (_hand_over(ctx, dx), *_match_params(_deliver(tg), ctx.params), None, None, None, None, None, None, None, None, None,
                None)


def nb1d(x, weights, biases, dilated, bn1=None, bn2=None, bn_mode=0, drop_scale=None, momentum=0.1,
         eps=1e-3):
    """non_bottleneck_1d.forward of the reference (train/erfnet_blocks.py:55-73) on the GPU.

    weights / biases: conv3x1_1, conv1x3_1, conv3x1_2, conv1x3_2.  bn1 / bn2: (weight, bias,
    running_mean, running_var) tuples, used when bn_mode != 0.  drop_scale: [N, C] factors
    (0 or 1/(1-p)) standing for nn.Dropout2d, or None.
    """
    g1 = be1 = rm1 = rv1 = g2 = be2 = rm2 = rv2 = None
    if bn_mode:
        g1, be1, rm1, rv1 = bn1
        g2, be2, rm2, rv2 = bn2
    tok = _ReluToken()
    return _tag_output(_apply(_Nb1d, x, weights[0], biases[0], weights[1], biases[1], weights[2], biases[2],
                                   weights[3], biases[3], g1, be1, g2, be2, rm1, rv1, rm2, rv2,
                                   drop_scale, dilated, bn_mode, momentum, eps, tok), tok)


# The backward of a run of blocks batches the weight-packing and slab-reduction launches of its blocks
# (erf_batch_begin / erf_batch_pack / erf_batch_end) unless ERFNET_NO_BATCH_BWD=1 (A/B runs); the results
# are bit-identical either way.
_BATCH_BWD = os.environ.get("ERFNET_NO_BATCH_BWD", "0") in ("", "0")


def set_batch_backward(enabled):
    """Turn the batched backward of :func:`nb1d_chain` on or off; returns the previous setting."""
    global _BATCH_BWD
    prev, _BATCH_BWD = _BATCH_BWD, bool(enabled)
    return prev


# The ReLU masks of a run of blocks travel to its backward as bit tensors (sign_bits of
# erf_nb1d_block_t) unless ERFNET_NO_SIGN_BITS=1 (A/B runs); results are bit-identical either way.
_SIGN_BITS = os.environ.get("ERFNET_NO_SIGN_BITS", "0") in ("", "0")


def set_sign_bits(enabled):
    """Turn the bit-tensor ReLU masks of :func:`nb1d_chain` on or off; returns the previous setting."""
    global _SIGN_BITS
    prev, _SIGN_BITS = _SIGN_BITS, bool(enabled)
    return prev


class _Nb1dChain(torch.autograd.Function):
    """A run of non_bottleneck_1d blocks as one autograd node: the forward hands the whole run to
    erf_nb1d_chain_forward, the backward walks the blocks in reverse through erf_nb1d_backward and
    passes the ReLU mask of each block's input gradient along (the run knows its own wiring)."""

    PER_BLOCK = 12      # w0 b0 w1 b1 w2 b2 w3 b3 g1 be1 g2 be2

    @staticmethod
    def forward(ctx, x, meta, out_token, *flat):
        lib = _lib.lib()
        ctx.in_token = _input_token(x)
        ctx.out_token = out_token
        x = _nhwc(x)
        n, c, h, wd = x.shape
        nb = len(meta)
        need = out_token.grad_on and any(ctx.needs_input_grad)
        blocks = (_lib.Nb1dBlockStruct * nb)()
        keep = []           # python objects that must outlive the call (ctypes arrays, BN structs)
        per_block = []
        for i, m in enumerate(meta):
            w0, b0, w1, b1, w2, b2, w3, b3, g1, be1, g2, be2 = flat[i * 12:(i + 1) * 12]
            mode = m["mode"]
            if need:
                ws_ = [_param(t) for t in (w0, w1, w2, w3)]
                bs_ = [_param(t) for t in (b0, b1, b2, b3)]
            else:       # inference: only the pointers are used, no detached views are created
                ws_ = [_plain(t) for t in (w0, w1, w2, w3)]
                bs_ = [_plain(t) for t in (b0, b1, b2, b3)]
            bn1 = bn2 = _NO_BN
            if mode:
                bn1 = _Bn(mode, g1, be1, m["rm1"], m["rv1"], m["momentum"], m["eps"], need)
                bn2 = _Bn(mode, g2, be2, m["rm2"], m["rv2"], m["momentum"], m["eps"], need)
            ds = m["drop_scale"]
            ds = None if ds is None else ds.to(torch.float32).contiguous()
            y = torch.empty_like(x)
            save = None
            if need:
                save = [torch.empty_like(x), torch.empty_like(x) if mode else None, torch.empty_like(x),
                        torch.empty_like(x), torch.empty_like(x) if mode else None]
            B = blocks[i]
            for k in range(4):
                B.w[k] = ws_[k].data_ptr()
                B.b[k] = bs_[k].data_ptr()
            if mode:
                B.bn1 = C.pointer(bn1.struct)
                B.bn2 = C.pointer(bn2.struct)
            B.drop_scale = _p(ds)
            B.dil = int(m["dil"])
            B.y = y.data_ptr()
            if save is not None:
                arr = _arr(save)
                keep.append(arr)
                B.save = C.cast(arr, C.POINTER(vp))
            bits = None
            if need and _SIGN_BITS and c % 32 == 0:
                # the ReLU masks of t1, t2, t3, y as bits (1/32 of an activation each): the library
                # fills those it can and says which; the backward's masked launches read them
                bits = torch.empty((4, n * h * wd * (c // 32)), dtype=torch.int32, device=x.device)
                barr = (vp * 4)(*[bits[k].data_ptr() for k in range(4)])
                flag = C.c_uint(0)
                keep += [barr, flag]
                B.sign_bits = C.cast(barr, C.POINTER(vp))
                B.sign_bits_written = C.pointer(flag)
                bits = (bits, flag)
            keep += [bn1, bn2, ws_, bs_, ds]
            per_block.append((y, save, ws_, bn1, bn2, ds, bits))
        ws = _workspace(x, n * h * wd, c)
        with _dev_ctx(x):
            _lib.check(_fn("erf_nb1d_chain_forward", x)(_p(x), blocks, nb, n, c, h, wd, _p(ws), ws.numel(),
                                                        _stream(x)))
        if need:
            tensors = [x]
            ctx.bits_written = []
            for y, save, ws_, bn1, bn2, ds, bits in per_block:
                written = bits[1].value if bits is not None else 0
                ctx.bits_written.append(written)
                tensors += [y, *save, *ws_, bn1.gamma, bn2.gamma, bn1.save_mean, bn1.save_invstd,
                            bn2.save_mean, bn2.save_invstd, ds, bits[0] if written else None]
            ctx.save_for_backward(*tensors)
            ctx.cfg = [(int(m["dil"]), m["mode"], m["momentum"], m["eps"]) for m in meta]
            ctx.params = flat
        return per_block[-1][0]

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        saved = ctx.saved_tensors
        x0 = saved[0]
        n, c, h, wd = x0.shape
        nb = len(ctx.cfg)
        per = 18            # tensors saved per block
        gy = _nhwc(gy, x0.dtype)
        last_flags = _bwd_flags(ctx, gy)
        grads = [None] * (nb * 12)
        batched = _BATCH_BWD and nb > 1
        ws = _workspace(x0, n * h * wd, c, nb if batched else 1)
        ws_slice = _workspace_slice(n * h * wd, c) if batched else ws.numel()
        calls, delivered = [], []
        for i in reversed(range(nb)):
            (y, t1, p2, t2, t3, p4, w0, w1, w2, w3, g1, g2, sm1, si1, sm2, si2, ds, bits) = saved[1 + i * per:1 + (i + 1) * per]
            x = saved[1 + (i - 1) * per] if i else x0
            # mask bits of t1, t2, t3 (this block's) and of x (the previous block's y), where written
            sb = [bits[k] if bits is not None and ctx.bits_written[i] >> k & 1 else None for k in range(3)]
            xbits = saved[(i - 1) * per + per] if i else None
            sb.append(xbits[3] if xbits is not None and ctx.bits_written[i - 1] >> 3 & 1 else None)
            dil, mode, momentum, eps = ctx.cfg[i]
            bn1 = _Bn(mode, g1, None, None, None, momentum, eps, False, saved=(sm1, si1))
            bn2 = _Bn(mode, g2, None, None, None, momentum, eps, False, saved=(sm2, si2))
            dx = torch.empty_like(x)
            cvec = (c,) if mode else None
            params = ctx.params[i * 12:(i + 1) * 12]
            tg = _grad_targets(params, (w0.shape, (c,), w1.shape, (c,), w2.shape, (c,), w3.shape, (c,),
                                        cvec, cvec, cvec, cvec), x.device)
            dws = [tg[k][0] for k in (0, 2, 4, 6)]
            dbs = [tg[k][0] for k in (1, 3, 5, 7)]
            dgs, dbts = [tg[8][0], tg[10][0]], [tg[9][0], tg[11][0]]
            # inside the run every block masks the gradient it hands to its producer, which then
            # skips its own masking pass; the ends follow the neighbours outside the run
            if i == nb - 1:
                flags = last_flags & _lib.BWD_GY_PREMASKED
            else:
                flags = _lib.BWD_GY_PREMASKED
            if i > 0 or ctx.in_token is not None:
                flags |= _lib.BWD_MASK_DX
            # every block of a batched run works in a workspace slice of its own
            wsp = vp(ws.data_ptr() + (nb - 1 - i) * ws_slice) if batched else _p(ws)
            if any(b is not None for b in sb):
                fn = _fn("erf_nb1d_backward_bits", x)
                args = (_p(x), _arr([w0, w1, w2, w3]), bn1.ref(), bn2.ref(), _p(ds), dil, _p(y),
                        _arr([t1, p2, t2, t3, p4]), _arr(sb), _p(gy), _p(dx), _arr(dws), _arr(dbs), _arr(dgs),
                        _arr(dbts), n, c, h, wd, flags, wsp, ws_slice, _stream(x))
            else:
                fn = _fn("erf_nb1d_backward", x)
                args = (_p(x), _arr([w0, w1, w2, w3]), bn1.ref(), bn2.ref(), _p(ds), dil, _p(y),
                        _arr([t1, p2, t2, t3, p4]), _p(gy), _p(dx), _arr(dws), _arr(dbs), _arr(dgs), _arr(dbts),
                        n, c, h, wd, flags, wsp, ws_slice, _stream(x))
            if batched:
                calls.append((fn, args, (bn1, bn2, dx, gy)))      # (the tuple keeps the operands alive)
            else:
                with _dev_ctx(x):
                    _lib.check(fn(*args))
            delivered.append((i, tg, params))
            gy = dx
        if batched:
            # two passes over the same calls: plan and collect the weight packs, pack them in one launch,
            # run; the slab reductions of all blocks follow together
            with _dev_ctx(x0):
                _lib.check(lib.erf_batch_begin())
                try:
                    for fn, args, _ in calls:
                        _lib.check(fn(*args))
                    _lib.check(lib.erf_batch_pack(_stream(x0)))
                    for fn, args, _ in calls:
                        _lib.check(fn(*args))
                    _lib.check(lib.erf_batch_end(_stream(x0)))
                except BaseException:
                    lib.erf_batch_abort()
                    raise
        for i, tg, params in delivered:
            grads[i * 12:(i + 1) * 12] = _match_params(_deliver(tg), params)
        return (_hand_over(ctx, gy), None, None, *grads)


def nb1d_chain(x, blocks):
    """``for b in blocks: x = nb1d(x, **b)`` as one call (reference: the layer loops of
    train/net_classification.py:32-36, 42-45).  Each element of ``blocks`` is a dict with the
    arguments of :func:`nb1d`: weights, biases, dilated, and optionally bn1, bn2, bn_mode,
    drop_scale, momentum, eps."""
    meta, flat = [], []
    for b in blocks:
        mode = int(b.get("bn_mode", 0))
        g1 = be1 = rm1 = rv1 = g2 = be2 = rm2 = rv2 = None
        if mode:
            g1, be1, rm1, rv1 = b["bn1"]
            g2, be2, rm2, rv2 = b["bn2"]
        meta.append({"dil": b["dilated"], "mode": mode, "momentum": b.get("momentum", 0.1),
                     "eps": b.get("eps", 1e-3), "drop_scale": b.get("drop_scale"),
                     "rm1": rm1, "rv1": rv1, "rm2": rm2, "rv2": rv2})
        w, bi = b["weights"], b["biases"]
        flat += [w[0], bi[0], w[1], bi[1], w[2], bi[2], w[3], bi[3], g1, be1, g2, be2]
    tok = _ReluToken()
    return _tag_output(_apply(_Nb1dChain, x, meta, tok, *flat), tok)


def conv_pair_pack(w_first, w_second):
    """Repack the [C,C,3,1] / [C,C,1,3] weights of two chained 1-D convolutions for
    :func:`conv_pair` (erf_conv_pair_pack); do it once per set of weights."""
    c = w_first.shape[0]
    wa, wb = _param(w_first), _param(w_second)
    packed = torch.empty(2 * 3 * c * c, dtype=torch.float32, device=wa.device)
    with _dev_ctx(wa):
        _lib.check(_lib.lib().erf_conv_pair_pack(_p(wa), _p(wb), c, _p(packed), _stream(wa)))
    return packed


def conv_pair(x, packed, b_first, b_second, first_axis, dil_first=1, dil_second=1, relu_first=True,
              relu_second=True, residual=None, keep_mid=False):
    """Two chained 3-tap convolutions in one launch, inference only (no autograd graph):
    ``mid = act(conv_first(x)); y = act(conv_second(mid) [+ residual])`` — the conv/relu/conv runs
    of non_bottleneck_1d.forward (reference train/erfnet_blocks.py:57-66).  first_axis 1: (1,3) then
    (3,1) kernels; 0: (3,1) then (1,3).  Returns y, or (y, mid) with keep_mid.  Raises when the
    image geometry has no fused kernel."""
    x = _nhwc(x.detach())
    if x.dtype != torch.float32:
        raise TypeError("conv_pair takes fp32 tensors (bf16: conv_pair_bf16)")
    n, c, h, wd = x.shape
    y = torch.empty_like(x)
    mid = torch.empty_like(x) if keep_mid else None
    res = None if residual is None else _nhwc(residual.detach())
    with _dev_ctx(x):
        _lib.check(_lib.lib().erf_conv_pair_forward(_p(x), _p(packed), _p(_param(b_first)), _p(_param(b_second)),
                                                    int(first_axis), int(dil_first), int(dil_second),
                                                    int(bool(relu_first)), int(bool(relu_second)), _p(res),
                                                    _p(mid), _p(y), n, c, h, wd, _stream(x)))
    return (y, mid) if keep_mid else y


def conv_line_pack_split(w):
    """Split the [C,C,3,1] / [C,C,1,3] weights of a 1-D convolution into three bf16 planes laid out
    as matrix fragments for :func:`conv_line_split` (erf_conv_line_pack_split)."""
    c = w.shape[0]
    wq = _param(w)
    packed = torch.empty(3 * 3 * c * c, dtype=torch.bfloat16, device=wq.device)
    with _dev_ctx(wq):
        _lib.check(_lib.lib().erf_conv_line_pack_split(_p(wq), c, _p(packed), _stream(wq)))
    return packed


def conv_line_split(x, packed, bias, axis, dil=1, relu=True, residual=None, mask=None):
    """One 3-tap convolution of non_bottleneck_1d.forward (reference train/erfnet_blocks.py:57-66) on
    fp32 tensors, computed with split-bf16 operands on the bf16 matrix pipe; inference only (no
    autograd graph).  axis 0: kernel (3,1); 1: kernel (1,3).
    ``y = where(mask > 0, act(conv(x) + bias [+ residual]), 0)``."""
    x = _nhwc(x.detach())
    if x.dtype != torch.float32:
        raise TypeError("conv_line_split takes fp32 tensors")
    n, c, h, wd = x.shape
    y = torch.empty_like(x)
    res = None if residual is None else _nhwc(residual.detach())
    mk = None if mask is None else _nhwc(mask.detach())
    with _dev_ctx(x):
        _lib.check(_lib.lib().erf_conv_line_forward_split(_p(x), _p(packed), _p(_param(bias)), int(axis), int(dil),
                                                          int(bool(relu)), _p(res), _p(mk), _p(y), n, c, h, wd,
                                                          _stream(x)))
    return y


# ------------------------------------------------------------------------------ UpsamplerBlock
class _Upsampler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rm, rv, mode, momentum, eps, out_token):
        lib = _lib.lib()
        ctx.in_token = _input_token(x)
        ctx.out_token = out_token
        x = _nhwc(x)
        n, cin, h, wd = x.shape
        cout = w.shape[1]
        need = out_token.grad_on and any(ctx.needs_input_grad)
        bn = _Bn(mode, gamma, beta, rm, rv, momentum, eps, need)
        wq, bq = _param(w), _param(b)
        y = _empty_nhwc(n, cout, 2 * h, 2 * wd, x)
        pre = torch.empty_like(y) if (mode and need) else None
        ws = _workspace(x, n * 4 * h * wd, max(cin, cout))
        with _dev_ctx(x):
            _lib.check(_fn("erf_upsampler_forward", x)(_p(x), _p(wq), _p(bq), bn.ref(), _p(y), _p(pre), n, cin,
                                                 h, wd, cout, _p(ws), ws.numel(), _stream(x)))
        if need:
            ctx.save_for_backward(x, wq, y, pre, bn.gamma, bn.save_mean, bn.save_invstd)
            ctx.cfg = (mode, momentum, eps)
            ctx.params = (w, b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        x, wq, y, pre, gamma, smean, sinv = ctx.saved_tensors
        mode, momentum, eps = ctx.cfg
        n, cin, h, wd = x.shape
        cout = y.shape[1]
        gy = _nhwc(gy, y.dtype)
        bn = _Bn(mode, gamma, None, None, None, momentum, eps, False, saved=(smean, sinv))
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        cvec = (cout,) if mode else None
        tg = _grad_targets(ctx.params, (wq.shape, (cout,), cvec, cvec), x.device)
        dw, db, dg, dbt = (t[0] for t in tg)
        ws = _workspace(x, n * 4 * h * wd, max(cin, cout))
        flags = _bwd_flags(ctx, gy)
        if dx is None:
            flags &= _lib.BWD_GY_PREMASKED
        with _dev_ctx(x):
            _lib.check(_fn("erf_upsampler_backward", x)(_p(x), _p(wq), bn.ref(), _p(y), _p(pre), _p(gy), _p(dx),
                                                  _p(dw), _p(db), _p(dg), _p(dbt), n, cin, h, wd, cout,
                                                  flags, _p(ws), ws.numel(), _stream(x)))
        dw, db, dg, dbt = _match_params(_deliver(tg), ctx.params)
        return _hand_over(ctx, dx), dw, db, dg, dbt, None, None, None, None, None, None


def upsampler(x, w, b, bn_weight=None, bn_bias=None, running_mean=None, running_var=None, bn_mode=0,
              momentum=0.1, eps=1e-3):
    """UpsamplerBlock.forward of the reference (train/erfnet_blocks.py:84-88) on the GPU."""
    tok = _ReluToken()
    return _tag_output(_apply(_Upsampler, x, w, b, bn_weight if bn_mode else None,
                                        bn_bias if bn_mode else None, running_mean, running_var, bn_mode,
                                        momentum, eps, tok), tok)


# ------------------------------------------------------------------------------ output ConvTranspose2d
# Loss hand-over.  The logits are the largest tensor of the network and their gradient would be
# written by the loss and read twice by the output convolution's backward.  While autograd records,
# the logits carry a token; CrossEntropyLoss2d's backward deposits what defines its gradient
# (targets, log-sum-exp, label count, upstream scalar) in the token and hands autograd a stride-0
# view of a zero scalar instead of the gradient; the output convolution's backward finds the deposit
# and runs one fused kernel (erf_outconv_ce_backward).  Because the stand-in is zero, gradients that
# other consumers of the logits add to it (the mean-teacher consistency loss, train/main.py:353)
# stay correct: the convolution then sees a tensor that is not the stand-in, materialises the
# deposited loss gradients and takes the ordinary path.  What cannot see the loss gradient is
# code that asks autograd for the gradient of the logits themselves (torch.autograd.grad(loss,
# logits), logits.retain_grad() after the loss was built): such a pass ends with the deposit still
# in the token, which raises (and withdraws the deposit, so that a later pass is not doubled);
# call set_loss_handover(False) for that use.
_LOSS_HANDOVER = True


def set_loss_handover(enabled):
    """Switch the loss hand-over (see above) on or off; returns the previous setting."""
    global _LOSS_HANDOVER
    old, _LOSS_HANDOVER = _LOSS_HANDOVER, bool(enabled)
    return old


# Deferred logits.  In the reference's training loop the logits have one reader, the loss
# (train/main.py:327,333: outputs = model(inputs); loss = criterion(outputs, targets)), and with the
# loss hand-over even the backward pass recomputes them in registers — writing 4*K bytes per output
# pixel only to read them once is the single largest transfer of the step.  So, while autograd
# records, the output convolution returns a tensor whose memory is allocated but not yet filled
# (_LazyLogits): CrossEntropyLoss2d recognises it and runs erf_outconv_ce_forward on the convolution's
# INPUT instead; anything else that touches the tensor — any torch function or Tensor method that is
# not a pure metadata query — first runs the convolution kernel into that memory, after which it is
# an ordinary tensor.  What cannot be intercepted is code that reads the memory without going through
# torch's Python API (DataParallel's gather of several replicas runs in C++): replicas therefore get
# ordinary logits (erfnet_blocks.OutputConvTranspose), and set_lazy_logits(False) turns it off.
_LAZY_LOGITS = os.environ.get("ERFNET_NO_LAZY_LOGITS", "0") in ("", "0")       # (the variable: A/B runs)


def set_lazy_logits(enabled):
    """Switch the deferred logits (see above) on or off; returns the previous setting."""
    global _LAZY_LOGITS
    old, _LAZY_LOGITS = _LAZY_LOGITS, bool(enabled)
    return old


def _tensor_args(args, kwargs):
    stack = list(args) + list(kwargs.values())
    while stack:
        a = stack.pop()
        if isinstance(a, torch.Tensor):
            yield a
        elif isinstance(a, (list, tuple)):
            stack.extend(a)
        elif isinstance(a, dict):
            stack.extend(a.values())


class _LazyLogits(torch.Tensor):
    """Logits of the output convolution whose kernel has not run yet (``_erf_lazy`` holds what it needs)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func not in _LAZY_METADATA:
            for a in _tensor_args(args, kwargs):
                if type(a) is _LazyLogits:
                    _materialize_logits(a)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _metadata_queries():
    t = torch.Tensor
    props = ("shape", "dtype", "device", "requires_grad", "grad_fn", "is_leaf", "ndim", "is_cuda", "layout",
             "_version", "retains_grad", "grad", "is_sparse", "is_quantized", "is_meta", "names", "output_nr")
    fns = {getattr(t, n).__get__ for n in props if hasattr(t, n) and hasattr(getattr(t, n), "__get__")}
    for n in ("size", "dim", "stride", "numel", "nelement", "is_contiguous", "element_size", "storage_offset",
              "is_floating_point", "is_complex", "get_device", "register_hook", "retain_grad", "__hash__",
              "is_shared", "is_pinned", "has_names", "__len__", "ndimension", "type"):
        if hasattr(t, n):
            fns.add(getattr(t, n))
    return fns


_LAZY_METADATA = _metadata_queries()


def _lazy_pending(t):
    """The deferred-convolution record of ``t`` if its kernel has not run yet, else None."""
    if type(t) is not _LazyLogits:
        return None
    rec = t.__dict__.get("_erf_lazy")
    return rec if (rec is not None and not rec["done"]) else None


def _materialize_logits(t):
    rec = _lazy_pending(t)
    if rec is None:
        return
    rec["done"] = True
    x, wb = rec["x"], rec["wb"]
    n, cin, h, wd = x.shape
    ncls = rec["ncls"]
    nw = cin * ncls * 4
    wq, bq = wb[:nw], (wb[nw:] if rec["has_bias"] else None)
    with torch._C.DisableTorchFunctionSubclass():
        ptr = torch.Tensor.data_ptr(t)
    lib = _lib.lib()
    with _dev_ctx(x):
        if x.dtype == torch.bfloat16:
            _lib.check(lib.erf_outconv_forward_bf16(_p(x), _p(wq), _p(bq), C.c_void_p(ptr), 0, n, cin, h, wd, ncls, 0,
                                                    _stream(x)))
        else:
            _lib.check(lib.erf_outconv_forward(_p(x), _p(wq), _p(bq), C.c_void_p(ptr), n, cin, h, wd, ncls, 0,
                                               _stream(x)))
    rec["x"] = rec["wb"] = None


class _LossToken:
    __slots__ = ("pending", "zero")

    _zeros = {}         # one read-only zero scalar per device (a fill launch per forward otherwise)

    def __init__(self, device):
        self.pending = []
        z = _LossToken._zeros.get(device)
        if z is None:
            z = _LossToken._zeros[device] = torch.zeros((), dtype=torch.float32, device=device)
        self.zero = z

    def is_stand_in(self, g):
        return g.data_ptr() == self.zero.data_ptr() and all(s == 0 for s in g.stride())


class _OutConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, loss_token, lazy):
        lib = _lib.lib()
        ctx.in_token = _input_token(x)
        x = _nhwc(x)
        n, cin, h, wd = x.shape
        ncls = w.shape[1]
        wq, bq = _param(w), _param(b)
        if lazy:
            # memory now, contents when somebody other than the loss asks for them (see _LazyLogits);
            # the parameters are copied: an optimiser step may come between now and then
            y = torch.empty((n, ncls, 2 * h, 2 * wd), dtype=torch.float32, device=x.device).as_subclass(_LazyLogits)
            wb = torch.cat([wq.reshape(-1), bq.reshape(-1)]) if bq is not None else wq.reshape(-1).clone()
            y._erf_lazy = dict(done=False, x=x, wb=wb, ncls=ncls, has_bias=bq is not None)
        else:
            y = _outconv_call(x, wq, bq, n, cin, h, wd, ncls, 0, False)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, wq, bq)
            ctx.has_bias = b is not None
            ctx.params = (w, b)
            ctx.loss_token = loss_token
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        x, wq, bq = ctx.saved_tensors
        n, cin, h, wd = x.shape
        ncls = wq.shape[1]
        tok = ctx.loss_token
        pending = []
        if tok is not None:
            pending, tok.pending = tok.pending, []
        fused = len(pending) == 1 and tok.is_stand_in(gy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        tg = _grad_targets(ctx.params, (wq.shape, (ncls,)), x.device)
        dw, db = tg[0][0], tg[1][0]
        ws = _workspace(x, n * h * wd, 16)
        flags = _lib.BWD_MASK_DX if (ctx.in_token is not None and dx is not None) else 0
        if fused:
            target, lse, stats, gl, ignore_index, _ = pending[0]
            with _dev_ctx(x):
                _lib.check(_fn("erf_outconv_ce_backward", x)(_p(x), _p(wq), _p(bq), _p(target), _p(lse), _p(stats),
                                                             _p(gl), _p(dx), _p(dw), _p(db), n, cin, h, wd, ncls,
                                                             ignore_index, flags, _p(ws), ws.numel(), _stream(x)))
        else:
            if pending:
                # other gradients were added to the stand-in (or several losses deposited): the
                # deposited loss gradients are materialised and summed with what autograd delivered
                total = None if tok.is_stand_in(gy) else gy.float()
                for target, lse, stats, gl, ignore_index, logits in pending:
                    _materialize_logits(logits)
                    g = _cross_entropy_grad(logits, target, lse, stats, gl, ignore_index)
                    total = g if total is None else total + g
                gy = total
            gy = gy.contiguous()
            if gy.dtype != torch.float32:
                gy = gy.float()
            with _dev_ctx(x):
                _lib.check(_fn("erf_outconv_backward", x)(_p(x), _p(wq), _p(gy), _p(dx), _p(dw), _p(db), n, cin, h,
                                                          wd, ncls, flags, _p(ws), ws.numel(), _stream(x)))
        dw, db = _match_params(_deliver(tg), ctx.params)
        return _hand_over(ctx, dx), dw, (db if ctx.has_bias else None), None, None


def _outconv_call(x, wq, bq, n, cin, h, wd, ncls, softmax, bf16_out):
    """Launch the output transposed convolution; bf16 activations give bf16 or fp32 logits."""
    lib = _lib.lib()
    if x.dtype == torch.bfloat16:
        y = torch.empty((n, ncls, 2 * h, 2 * wd), dtype=torch.bfloat16 if bf16_out else torch.float32,
                        device=x.device)
        with _dev_ctx(x):
            _lib.check(lib.erf_outconv_forward_bf16(_p(x), _p(wq), _p(bq), _p(y), int(bool(bf16_out)), n, cin, h,
                                                    wd, ncls, int(softmax), _stream(x)))
        return y
    y = torch.empty((n, ncls, 2 * h, 2 * wd), dtype=torch.float32, device=x.device)
    with _dev_ctx(x):
        _lib.check(lib.erf_outconv_forward(_p(x), _p(wq), _p(bq), _p(y), n, cin, h, wd, ncls, int(softmax),
                                           _stream(x)))
    return y


def outconv(x, w, b, allow_lazy=True):
    """nn.ConvTranspose2d(Cin, ncls, 2, stride=2) (train/net_classification.py:56,64) on the GPU.
    Returns NCHW-contiguous logits.  bf16 activations: the logits are fp32 while an autograd graph is
    being built (they feed the loss), bf16 under torch.no_grad() (half the bytes of the largest
    tensor of the network).  While an autograd graph is being built the logits may be deferred
    (_LazyLogits above) unless ``allow_lazy`` is False."""
    if torch.is_grad_enabled():
        tok = _LossToken(x.device) if _LOSS_HANDOVER else None
        lazy = bool(tok is not None and _LAZY_LOGITS and allow_lazy and w.shape[1] <= 32 and x.shape[1] == 16 and
                    (x.requires_grad or w.requires_grad))
        y = _OutConv.apply(x, w, b, tok, lazy)
        if tok is not None and y.requires_grad:
            y._erf_loss_token = tok
        elif lazy:
            _materialize_logits(y)
        return y
    x = _nhwc(x)
    n, cin, h, wd = x.shape
    return _outconv_call(x, _param(w), _param(b), n, cin, h, wd, w.shape[1], 0, x.dtype == torch.bfloat16)


def outconv_softmax(x, w, b):
    """SoftMaxConv.forward without dropout (train/erfnet_blocks.py:104-110), inference only:
    transposed convolution and softmax over the classes in one kernel."""
    lib = _lib.lib()
    x = _nhwc(x)
    n, cin, h, wd = x.shape
    ncls = w.shape[1]
    return _outconv_call(x, _param(w), _param(b), n, cin, h, wd, ncls, 1, x.dtype == torch.bfloat16)


def outconv_argmax(x, w, b):
    """``output_conv(x).max(1)[1].unsqueeze(1)`` (reference eval/eval_iou.py:92) in one kernel,
    inference only: int64 class indices [N,1,2H,2W]; the logits never reach memory."""
    lib = _lib.lib()
    x = _nhwc(x.detach())
    n, cin, h, wd = x.shape
    ncls = w.shape[1]
    wq, bq = _param(w), _param(b)
    labels = torch.empty((n, 1, 2 * h, 2 * wd), dtype=torch.int64, device=x.device)
    with _dev_ctx(x):
        _lib.check(_fn("erf_outconv_argmax", x)(_p(x), _p(wq), _p(bq), _p(labels), n, cin, h, wd, ncls, _stream(x)))
    return labels


# ------------------------------------------------------------------------------ plain Conv2d (1x1 / 3x3)
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        lib = _lib.lib()
        x = _nhwc(x)
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        wq, bq = _param(w), _param(b)
        y = _empty_nhwc(n, cout, h, wd, x)
        ws = _workspace(x, n * h * wd, max(cin, cout))
        with _dev_ctx(x):
            _lib.check(_fn("erf_conv2d_forward", x)(_p(x), _p(wq), _p(bq), _p(y), n, cin, h, wd, cout, k,
                                                    int(relu), _p(ws), ws.numel(), _stream(x)))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, wq, y if relu else None)
            ctx.cfg = (k, int(relu), b is not None)
            ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        x, wq, y = ctx.saved_tensors
        k, relu, has_bias = ctx.cfg
        n, cin, h, wd = x.shape
        cout = wq.shape[0]
        gy = _nhwc(gy, x.dtype)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        tg = _grad_targets(ctx.params, (wq.shape, (cout,)), x.device)
        dw, db = tg[0][0], tg[1][0]
        ws = _workspace(x, n * h * wd, max(cin, cout))
        with _dev_ctx(x):
            _lib.check(_fn("erf_conv2d_backward", x)(_p(x), _p(wq), _p(y), _p(gy), _p(dx), _p(dw), _p(db), n, cin,
                                                     h, wd, cout, k, relu, _p(ws), ws.numel(), _stream(x)))
        dw, db = _match_params(_deliver(tg), ctx.params)
        return dx, dw, (db if has_bias else None), None


def conv2d(x, w, b, relu=False):
    """nn.Conv2d with a 1x1 or 3x3 kernel, stride 1, 'same' padding (Encoder.output_conv,
    eval/erfnet.py:85; LadderBlock.conv, train/erfnet_blocks.py:145), optionally fused with ReLU."""
    return _apply(_Conv2d, x, w, b, relu)


# ------------------------------------------------------------------------------ CrossEntropyLoss2d
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lib = _lib.lib()
        if logits.dtype != torch.float32:
            raise TypeError(f"cross_entropy: the loss reads fp32 logits, got {logits.dtype}")
        if target.dtype != torch.int64:
            raise TypeError(f"cross_entropy: target must be int64 class indices, got {target.dtype}")
        n, k, h, wd = logits.shape
        if tuple(target.shape) != (n, h, wd):
            raise ValueError(f"cross_entropy: target shape {tuple(target.shape)} does not match logits "
                             f"{tuple(logits.shape)}")
        tg = target.contiguous()
        dev = logits.device
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        lse = torch.empty((n, h, wd), dtype=torch.float32, device=dev)
        # With the reference's ignore_index (-1: exactly the negative labels are unlabelled) the same
        # pass also counts what its accuracy metrics need; they find the counts on the logits.
        counts = torch.empty((2, k), dtype=torch.int64, device=dev) if (k <= 32 and int(ignore_index) == -1) \
            else None
        rec = _lazy_pending(logits)
        tok = logits.__dict__.get("_erf_loss_token") if rec is not None else None
        if rec is not None and tok is not None and _LOSS_HANDOVER and ctx.needs_input_grad[0] and \
                tg.data_ptr() % 16 == 0:
            # deferred logits whose gradient will go through the hand-over: loss, log-sum-exp and counts
            # straight from the output convolution's input (erf_outconv_ce_forward)
            xin, wb = rec["x"], rec["wb"]
            _, cin, hi, wi = xin.shape
            nw = cin * k * 4
            wq, bq = wb[:nw], (wb[nw:] if rec["has_bias"] else None)
            ws = _workspace(xin, n * hi * wi, 2 * k + 2)
            with _dev_ctx(xin):
                _lib.check(_fn("erf_outconv_ce_forward", xin)(_p(xin), _p(wq), _p(bq), _p(tg), _p(stats), _p(lse),
                                                              _p(counts), n, cin, hi, wi, k, int(ignore_index),
                                                              _p(ws), ws.numel(), _stream(xin)))
            if counts is not None:
                logits._erf_acc_cache = (target.data_ptr(), target._version, tuple(target.shape),
                                         torch.Tensor._version.__get__(logits), counts)
            ctx.save_for_backward(tg, lse, stats)
            ctx.lazy_logits = logits
            ctx.ignore_index = int(ignore_index)
            ctx.loss_token = tok
            return stats[0]
        _materialize_logits(logits)
        x = logits.contiguous()
        ws = _workspace(x, n * h * wd, 1)
        with _dev_ctx(x):
            if counts is None:
                _lib.check(lib.erf_cross_entropy_forward(_p(x), _p(tg), _p(stats), _p(lse), n, k, h, wd,
                                                         int(ignore_index), _p(ws), ws.numel(), _stream(x)))
            else:
                _lib.check(lib.erf_cross_entropy_forward_counts(_p(x), _p(tg), _p(stats), _p(lse), _p(counts), n, k,
                                                                h, wd, int(ignore_index), _p(ws), ws.numel(),
                                                                _stream(x)))
                logits._erf_acc_cache = (target.data_ptr(), target._version, tuple(target.shape), logits._version,
                                         counts)
        ctx.lazy_logits = None
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, tg, lse, stats)
            ctx.ignore_index = int(ignore_index)
            tok = getattr(logits, "_erf_loss_token", None) if _LOSS_HANDOVER else None
            # the hand-over needs the very tensor the output convolution produced
            ctx.loss_token = tok if (tok is not None and x.data_ptr() == logits.data_ptr()
                                     and not logits.retains_grad) else None
        return stats[0]

    @staticmethod
    def backward(ctx, gy):
        if ctx.lazy_logits is not None:
            tg, lse, stats = ctx.saved_tensors
            x = ctx.lazy_logits         # (filled only if the gradient has to be materialised after all)
        else:
            x, tg, lse, stats = ctx.saved_tensors
        gy = gy.to(torch.float32).reshape(1).contiguous()
        tok = ctx.loss_token
        if tok is not None and _LOSS_HANDOVER:
            entry = (tg, lse, stats, gy, ctx.ignore_index, x)
            tok.pending.append(entry)

            def settle():
                # End of this backward pass.  A deposit nobody collected means the output convolution's
                # backward did not run in it — the caller asked autograd for the gradient of the logits
                # themselves and was handed the zero stand-in.  Left in place, the deposit would also be
                # added to the next pass's gradient.  Withdraw it and fail loudly.
                for i, e in enumerate(tok.pending):
                    if e is entry:
                        del tok.pending[i]
                        raise RuntimeError(
                            "erfnet_pytorch_amd: the gradient of the logits was requested from autograd while the "
                            "loss hand-over is on (CrossEntropyLoss2d hands its gradient straight to the output "
                            "convolution's backward); call functional.set_loss_handover(False) before building "
                            "the graph to get d(loss)/d(logits) itself")
            torch.autograd.Variable._execution_engine.queue_callback(settle)
            return tok.zero.expand(x.shape), None, None
        _materialize_logits(x)
        return _cross_entropy_grad(x, tg, lse, stats, gy, ctx.ignore_index), None, None


def _cross_entropy_grad(x, tg, lse, stats, gy, ignore_index):
    lib = _lib.lib()
    n, k, h, wd = x.shape
    dx = torch.empty_like(x)
    with _dev_ctx(x):
        _lib.check(lib.erf_cross_entropy_backward(_p(x), _p(tg), _p(lse), _p(stats), _p(gy), _p(dx),
                                                  n, k, h, wd, ignore_index, _stream(x)))
    return dx


def cross_entropy(logits, target, ignore_index=-1):
    """nn.CrossEntropyLoss(ignore_index)(logits, target) for [N,K,H,W] logits and [N,H,W] int64
    targets (the reference's CrossEntropyLoss2d, train/evaluation_functions.py:6-14) on the GPU."""
    return _CrossEntropy.apply(logits, target, ignore_index)


# ------------------------------------------------------------------------------ masked regression losses
ML_L1, ML_LL, ML_L1CP, ML_LLCP = 0, 1, 2, 3


def _f32c(t, what):
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: fp32 tensors only, got {t.dtype}")
    return t.contiguous()


class _MaskedLoss(torch.autograd.Function):
    """a: [N,K,H,W]; b: [N,H,W] (ML_LL), [K] values (class kinds) or None; c: [K] values or None;
    target: [N,H,W] float, -1 = no label."""

    @staticmethod
    def forward(ctx, kind, eps, a, b, c, target):
        lib = _lib.lib()
        n, k, h, wd = a.shape
        a = _f32c(a, "masked loss")
        b = None if b is None else _f32c(b, "masked loss")
        c = None if c is None else _f32c(c, "masked loss")
        tg = _f32c(target, "masked loss")
        if tg.numel() != n * h * wd:
            raise ValueError(f"masked loss: target shape {tuple(target.shape)} does not match {tuple(a.shape)}")
        want = {ML_L1: 0, ML_LL: n * h * wd, ML_L1CP: k, ML_LLCP: k}[kind]
        if (0 if b is None else b.numel()) != want or (0 if c is None else c.numel()) != (k if kind == ML_LLCP else 0):
            raise ValueError("masked loss: operand sizes do not match the kind")
        stats = torch.empty(2, dtype=torch.float32, device=a.device)
        ws = _workspace(a, n * h * wd, k)
        with _dev_ctx(a):
            _lib.check(lib.erf_masked_loss_forward(kind, _p(a), _p(b), _p(c), _p(tg), float(eps), _p(stats), n, k,
                                                   h, wd, _p(ws), ws.numel(), _stream(a)))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(a, b, c, tg, stats)
            ctx.cfg = (kind, float(eps))
        return stats[0]

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        a, b, c, tg, stats = ctx.saved_tensors
        kind, eps = ctx.cfg
        n, k, h, wd = a.shape
        gy = gy.to(torch.float32).reshape(1).contiguous()
        da = torch.empty_like(a)
        db = None if b is None else torch.empty_like(b)
        dc = None if c is None else torch.empty_like(c)
        ws = _workspace(a, n * h * wd, k)
        with _dev_ctx(a):
            _lib.check(lib.erf_masked_loss_backward(kind, _p(a), _p(b), _p(c), _p(tg), eps, _p(stats), _p(gy),
                                                    _p(da), _p(db), _p(dc), n, k, h, wd, _p(ws), ws.numel(),
                                                    _stream(a)))
        return None, None, da, db, dc, None


def masked_loss(kind, a, b, c, target, eps=0.0):
    """Mean over the labelled pixels (target != -1) of the per-pixel value of one of the reference's
    masked regression losses (train/evaluation_functions.py:32-97), forward and backward as one HIP
    kernel each.  a [N,K,H,W]; b, c: see _MaskedLoss."""
    return _MaskedLoss.apply(kind, eps, a, b, c, target)


class _MseMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.lib()
        if a.shape != b.shape:
            raise ValueError(f"mse: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        a, b = _f32c(a, "mse"), _f32c(b, "mse")
        stats = torch.empty(2, dtype=torch.float32, device=a.device)
        ws = _workspace(a, a.numel(), 1)
        with _dev_ctx(a):
            _lib.check(lib.erf_mse_mean_forward(_p(a), _p(b), a.numel(), _p(stats), _p(ws), ws.numel(), _stream(a)))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(a, b)
        return stats[0]

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        a, b = ctx.saved_tensors
        gy = gy.to(torch.float32).reshape(1).contiguous()
        da = torch.empty_like(a)
        with _dev_ctx(a):
            _lib.check(lib.erf_mse_mean_backward(_p(a), _p(b), _p(gy), _p(da), a.numel(), _stream(a)))
        # d/db = -d/da; the consistency loss's second operand is the detached teacher output
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


def mse_mean(a, b):
    """mean((a - b)^2) (the elementwise part of MSELossWeighted, train/evaluation_functions.py:122-131)."""
    return _MseMean.apply(a, b)
