"""Mean-teacher weight update of the reference's training loop (train/main.py:133-142, used at
:390) as a multi-tensor kernel: a few launches for the whole model instead of three kernels and one
temporary per parameter tensor."""
import ctypes as C

import torch

from . import _lib

__all__ = ["copyWeightsToModelNoGrad", "copyWeightsToModelWithDiscount"]


def copyWeightsToModelNoGrad(model_source, model_target):
    """Freeze the target's parameters (train/main.py:133-138).  As in the reference, the values
    are left as they are — callers initialise the teacher from a checkpoint or by
    ``load_state_dict`` — only ``requires_grad`` is cleared."""
    for target in model_target.parameters():
        target.requires_grad = False


def copyWeightsToModelWithDiscount(model_source, model_target, discount_factor):
    """target = target * (1 - discount_factor) + discount_factor * source for every parameter
    (train/main.py:140-142), in place.  fp32 parameters on a GPU go through ``erf_ema_update`` (one
    launch per 32 tensors, bit-identical to the reference's three tensor operations); host tensors
    through torch's multi-tensor operations."""
    sources = [p.data for p in model_source.parameters()]
    targets = [p.data for p in model_target.parameters()]
    if len(sources) != len(targets):
        raise ValueError("the two models differ in their parameter lists")
    if not targets:
        return
    for s, t in zip(sources, targets):
        if s.shape != t.shape:
            raise ValueError(f"parameter shapes differ: {tuple(s.shape)} against {tuple(t.shape)}")
    if all(t.is_cuda and s.is_cuda and s.device == t.device and t.dtype == torch.float32 and
           s.dtype == torch.float32 and t.is_contiguous() and s.is_contiguous()
           for s, t in zip(sources, targets)):
        by_dev = {}
        for s, t in zip(sources, targets):
            by_dev.setdefault(t.device, []).append((s, t))
        for dev, pairs in by_dev.items():
            n = len(pairs)
            tp = (C.c_void_p * n)(*[t.data_ptr() for _, t in pairs])
            sp = (C.c_void_p * n)(*[s.data_ptr() for s, _ in pairs])
            ne = (C.c_longlong * n)(*[t.numel() for _, t in pairs])
            _lib.check(_lib.lib().erf_ema_update(n, tp, sp, ne, float(discount_factor),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return
    # Any other storage type or device: torch's multi-tensor operations on the parameters themselves
    # (not on their .data aliases, whose in-place updates leave the parameters' version counters
    # untouched — the fp32 copies that functional.py keeps of non-fp32 parameters are refreshed by
    # that counter), and the copies are dropped explicitly as well.
    params = list(model_target.parameters())
    with torch.no_grad():
        torch._foreach_mul_(params, 1.0 - discount_factor)
        torch._foreach_add_(params, sources, alpha=discount_factor)
    from . import functional
    functional.invalidate_parameter_copies()
