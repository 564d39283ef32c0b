"""Mean-teacher weight update of the reference's training loop (train/main.py:133-142, used at :390) as multi-tensor operations: two launches for the whole model instead of three kernels and
one temporary per parameter tensor."""
import torch

__all__ = ["copyWeightsToModelNoGrad", "copyWeightsToModelWithDiscount"]


def copyWeightsToModelNoGrad(model_source, model_target):
    """Freeze the target's parameters (train/main.py:133-138).  As in the reference, the values
    are left as they are — callers initialise the teacher from a checkpoint or by
    ``load_state_dict`` — only ``requires_grad`` is cleared."""
    for target in model_target.parameters():
        target.requires_grad = False


def copyWeightsToModelWithDiscount(model_source, model_target, discount_factor):
    """target = target * (1 - discount_factor) + discount_factor * source for every parameter
    (train/main.py:140-142), in place."""
    sources = [p.data for p in model_source.parameters()]
    targets = [p.data for p in model_target.parameters()]
    if len(sources) != len(targets):
        raise ValueError("the two models differ in their parameter lists")
    if not targets:
        return
    with torch.no_grad():
        torch._foreach_mul_(targets, 1.0 - discount_factor)
        torch._foreach_add_(targets, sources, alpha=discount_factor)
