"""numpy restatement of the parameter updates the reference's training loops perform.

TEST INFRASTRUCTURE ONLY (imported by ``tests/`` as the checker; the product package never imports it).

The algorithms live in a third-party dependency of the reference, PyTorch (the reference pins no
version; the fixtures in tests/golden/optim.npz were generated with the torch of this image, 2.10):
  * ``adam_step``  — torch.optim.Adam as called at train/main.py:213 (L2 weight decay folded into the
    gradient, bias corrections formed in double on the host, no amsgrad), stepped at train/main.py:381
  * ``sgd_step``   — torch.optim.SGD as called at imagenet/main.py:83-85 (momentum, dampening 0, no
    Nesterov; the momentum buffer starts as the first gradient), stepped at imagenet/main.py:187
  * ``ema_update`` — copyWeightsToModelWithDiscount, train/main.py:140-142
Everything is float32 arithmetic in the order torch's single-tensor code performs it; the functions
update their array arguments in place.
"""
import math

import numpy as np

f32 = np.float32


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """One update; ``step`` is the 1-based count of this update."""
    if weight_decay != 0:
        g = g + f32(weight_decay) * p
    m += f32(1.0 - beta1) * (g - m)                       # exp_avg.lerp_(grad, 1 - beta1)
    v *= f32(beta2)
    v += (f32(1.0 - beta2) * g) * g                       # addcmul_(grad, grad, value=1 - beta2)
    step_size = f32(lr / (1.0 - beta1 ** step))
    bc2_sqrt = f32(math.sqrt(1.0 - beta2 ** step))
    denom = np.sqrt(v) / bc2_sqrt + f32(eps)
    p -= (step_size * m) / denom                          # addcdiv_(exp_avg, denom, value=-step_size)


def sgd_step(p, g, buf, first, lr, momentum, weight_decay):
    """One update; ``first``: the momentum buffer does not exist yet (it becomes the gradient)."""
    if weight_decay != 0:
        g = g + f32(weight_decay) * p
    if momentum != 0:
        if first:
            buf[...] = g
        else:
            buf *= f32(momentum)
            buf += g
        g = buf
    p -= f32(lr) * g


def ema_update(target, source, discount):
    target[...] = target * f32(1.0 - discount) + f32(discount) * source
