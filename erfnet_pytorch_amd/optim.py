"""The optimisers of the reference's training loops on the HIP library's multi-tensor kernels.

``Adam`` takes the arguments of the ``torch.optim.Adam`` call in train/main.py:213
(``Adam(params, lr, betas, eps=..., weight_decay=...)``), ``SGD`` those of imagenet/main.py:83-85
(``SGD(params, lr, momentum=..., weight_decay=...)``).  Both are ``torch.optim.Optimizer`` subclasses:
``param_groups`` (learning-rate schedulers, imagenet/main.py:276-286 rewrites ``lr`` and
``weight_decay`` every epoch), ``zero_grad`` and ``state_dict`` / ``load_state_dict`` behave as in
torch, and the per-parameter state uses torch's key names (``step``, ``exp_avg``, ``exp_avg_sq``,
``momentum_buffer``), so a checkpoint written by the reference (train/main.py:232,467) resumes here.

One ``step()`` is one kernel launch per 32 parameter tensors (``erf_adam_step`` / ``erf_sgd_step``); on
the flat buffers of ``ddp.GradientReducer.flatten_parameters()`` that is a single launch for the model.
Parameters must be fp32 tensors on a GPU: there is no host path.
"""
import ctypes as C

import torch

from . import _lib

__all__ = ["Adam", "SGD"]


class _Tables:
    """Host pointer tables of one launch list (ctypes arrays, rebuilt when a pointer moves)."""

    def __init__(self, n_lists):
        self.key = None
        self.ptrs = [None] * n_lists
        self.numel = None

    def update(self, lists):
        key = tuple(t.data_ptr() for lst in lists for t in lst)
        if key != self.key:
            n = len(lists[0])
            for i, lst in enumerate(lists):
                self.ptrs[i] = (C.c_void_p * n)(*[t.data_ptr() for t in lst])
            self.numel = (C.c_longlong * n)(*[t.numel() for t in lists[0]])
            self.key = key
        return self.ptrs, self.numel


def _usable(p, what):
    if not p.is_cuda:
        raise RuntimeError(f"erfnet_pytorch_amd.optim: {what} lives on {p.device}; the update kernels "
                           "run on the GPU only (use torch.optim for host tensors)")
    if p.dtype != torch.float32:
        raise TypeError(f"erfnet_pytorch_amd.optim: {what} is {p.dtype}; parameters and gradients are fp32")
    if not p.is_contiguous():
        raise ValueError(f"erfnet_pytorch_amd.optim: {what} is not contiguous")


def _launch_groups(params):
    """Parameters with a gradient, grouped by device (one launch list per device)."""
    by_dev = {}
    for p in params:
        if p.grad is None:
            continue
        if p.grad.is_sparse:
            raise RuntimeError("erfnet_pytorch_amd.optim: sparse gradients are not supported")
        _usable(p, "a parameter")
        _usable(p.grad, "a gradient")
        by_dev.setdefault(p.device, []).append(p)
    return by_dev


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam`` with L2 weight decay (no amsgrad, no maximize) in one launch per 32 tensors."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("Adam: lr, eps and weight_decay must not be negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Adam: betas {betas} outside [0, 1)")
        # torch's remaining group keys at the values this optimiser implements, so that state dicts
        # travel between the two classes in both directions
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=False))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                raise NotImplementedError("erfnet_pytorch_amd.optim.Adam: amsgrad / maximize / decoupled weight "
                                          "decay are not implemented (the reference uses none of them)")
            for dev, ps in _launch_groups(group["params"]).items():
                for p in ps:
                    st = self.state[p]
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # parameters of a group that received their first gradient at different times carry
                # different step counts: one launch list per count
                by_step = {}
                for p in ps:
                    by_step.setdefault(int(self.state[p]["step"]), []).append(p)
                for si, (step, sub) in enumerate(by_step.items()):
                    lists = [sub, [p.grad for p in sub], [self.state[p]["exp_avg"] for p in sub],
                             [self.state[p]["exp_avg_sq"] for p in sub]]
                    for t in lists[2] + lists[3]:
                        _usable(t, "optimiser state")
                    tab = self._tables.setdefault((gi, dev, si), _Tables(4))
                    ptrs, numel = tab.update(lists)
                    b1, b2 = group["betas"]
                    _lib.check(lib.erf_adam_step(
                        len(sub), ptrs[0], ptrs[1], ptrs[2], ptrs[3], numel, float(group["lr"]), float(b1),
                        float(b2), float(group["eps"]), float(group["weight_decay"]), step + 1,
                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
                    for p in sub:
                        self.state[p]["step"] = torch.tensor(float(step + 1))
        return loss


class SGD(torch.optim.Optimizer):
    """``torch.optim.SGD`` with momentum and L2 weight decay (dampening 0, no Nesterov)."""

    def __init__(self, params, lr=1e-3, momentum=0, weight_decay=0):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("SGD: lr, momentum and weight_decay must not be negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                                      nesterov=False, maximize=False, foreach=None, differentiable=False,
                                      fused=None))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            if group.get("nesterov") or group.get("maximize") or group.get("dampening"):
                raise NotImplementedError("erfnet_pytorch_amd.optim.SGD: Nesterov momentum / dampening / maximize "
                                          "are not implemented (the reference uses none of them)")
            for dev, ps in _launch_groups(group["params"]).items():
                bufs = []
                for p in ps:
                    st = self.state[p]
                    if st.get("momentum_buffer") is None:
                        st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    _usable(st["momentum_buffer"], "optimiser state")
                    bufs.append(st["momentum_buffer"])
                tab = self._tables.setdefault((gi, dev), _Tables(3))
                ptrs, numel = tab.update([ps, [p.grad for p in ps], bufs])
                _lib.check(lib.erf_sgd_step(
                    len(ps), ptrs[0], ptrs[1], ptrs[2], numel, float(group["lr"]), float(group["momentum"]),
                    float(group["weight_decay"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss
