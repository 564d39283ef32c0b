"""Drop-in for the losses and accuracy metrics of ``train/evaluation_functions.py``.

``CrossEntropyLoss2d`` (train/evaluation_functions.py:6-14) is
``torch.nn.CrossEntropyLoss(ignore_index=-1)`` behind a module; ``train/main.py:186-188`` builds
it and ``:333,373`` call it on the classifier logits.  On GPU tensors the forward and backward run
as one HIP kernel each (erfnet_pytorch_amd/csrc/loss_kernels.h); CPU tensors go through torch.
"""
import numpy as np
import torch

from . import _lib
from . import functional as F_hip


class CrossEntropyLoss2d(torch.nn.Module):

    def __init__(self, weight=None):
        super().__init__()
        # the reference accepts `weight` and ignores it (train/evaluation_functions.py:8-11)
        self.loss = torch.nn.CrossEntropyLoss(ignore_index=-1)

    def forward(self, outputs, targets):
        if outputs.is_cuda and outputs.dim() == 4 and outputs.dtype == torch.float32:
            return F_hip.cross_entropy(outputs, targets, -1)
        return self.loss(outputs, targets)


def _gpu_f32(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


def _labelled_mean(values, targets):
    """CPU path of the masked losses: mean of `values` where the target carries a label."""
    return values[targets != -1.0].mean()


class L1LossMasked(torch.nn.Module):
    """mean |output - target| over the pixels whose target is not -1
    (train/evaluation_functions.py:89-97); outputs [N,1,H,W], targets [N,H,W]."""

    def forward(self, outputs, targets):
        if _gpu_f32(outputs, targets) and outputs.dim() == 4 and outputs.shape[1] == 1:
            return F_hip.masked_loss(F_hip.ML_L1, outputs, None, None, targets)
        return _labelled_mean((outputs.squeeze(1) - targets).abs(), targets)


class LogLikelihoodLossMasked(torch.nn.Module):
    """mean of log(2 pi s) + (cost - target)^2 / s, s = var + opt_eps, over the labelled pixels — the
    Gaussian negative log-likelihood without its factor 1/2 (train/evaluation_functions.py:71-87)."""

    def __init__(self, opt_eps=0.0):
        super().__init__()
        self.opt_eps = opt_eps

    def forward(self, output_cost, output_var, targets):
        if _gpu_f32(output_cost, output_var, targets) and output_cost.shape == output_var.shape == targets.shape:
            n = targets.shape[0] if targets.dim() > 1 else 1
            a = output_cost.reshape(n, 1, 1, -1)
            return F_hip.masked_loss(F_hip.ML_LL, a, output_var.reshape(n, 1, -1), None, targets.reshape(n, 1, -1),
                                     self.opt_eps)
        s = output_var + self.opt_eps
        return _labelled_mean(torch.log(2 * np.pi * s) + (output_cost - targets) ** 2 / s, targets)


def _class_operands(prob, *per_class):
    k = prob.shape[1]
    return all(t.numel() == k for t in per_class)


class L1LossClassProbMasked(torch.nn.Module):
    """mean over the labelled pixels of sum_k prob_k |cost_k - target|
    (train/evaluation_functions.py:32-47); prob [N,K,H,W], cost [1,K,1,1], targets [N,H,W]."""

    def forward(self, output_prob, output_cost, targets):
        if _gpu_f32(output_prob, output_cost, targets) and output_prob.dim() == 4 \
                and _class_operands(output_prob, output_cost) and output_prob.shape[1] <= 256:
            return F_hip.masked_loss(F_hip.ML_L1CP, output_prob, output_cost.reshape(-1), None, targets)
        t = targets.unsqueeze(1)
        per_pixel = ((output_cost - t).abs() * output_prob).sum(dim=1, keepdim=True)
        return _labelled_mean(per_pixel, t)


class LogLikelihoodLossClassProbMasked(torch.nn.Module):
    """mean over the labelled pixels of sum_k prob_k (log(2 pi s_k) + (cost_k - target)^2 / s_k),
    s_k = var_k + opt_eps (train/evaluation_functions.py:49-69)."""

    def __init__(self, opt_eps=0.0):
        super().__init__()
        self.opt_eps = opt_eps

    def forward(self, output_prob, output_cost, output_var, targets):
        if _gpu_f32(output_prob, output_cost, output_var, targets) and output_prob.dim() == 4 \
                and _class_operands(output_prob, output_cost, output_var) and output_prob.shape[1] <= 256:
            return F_hip.masked_loss(F_hip.ML_LLCP, output_prob, output_cost.reshape(-1), output_var.reshape(-1),
                                     targets, self.opt_eps)
        t = targets.unsqueeze(1)
        s = output_var + self.opt_eps
        per_class = torch.log(2 * np.pi * s) + (output_cost - t) ** 2 / s
        return _labelled_mean((per_class * output_prob).sum(dim=1, keepdim=True), t)


class L1Loss(torch.nn.Module):
    """nn.L1Loss with mean reduction behind a module (train/evaluation_functions.py:112-120)."""

    def forward(self, outputs, targets):
        return torch.nn.functional.l1_loss(outputs, targets)


class MSELossWeighted(torch.nn.Module):
    """mean((outputs - targets)^2 * scale_factor * weight): the mean-teacher consistency loss
    (train/evaluation_functions.py:122-131; train/main.py:191,341,353)."""

    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, outputs, targets, weight):
        if _gpu_f32(outputs, targets) and outputs.shape == targets.shape and outputs.numel() > 0:
            return F_hip.mse_mean(outputs, targets) * (self.scale_factor * weight)
        return ((outputs - targets) ** 2 * self.scale_factor * weight).mean()


def _accuracy_counts(prob, targets):
    """int64 [2][K] on the device: per label class, correctly classified pixels and labelled pixels
    (one pass of erf_accuracy_counts over the scores, no argmax tensor, no per-class loop; no pass
    at all when the loss has already been evaluated on the same tensors, as in train/main.py:327-347)."""
    cached = getattr(prob, "_erf_acc_cache", None)
    if cached is not None and cached[:4] == (targets.data_ptr(), targets._version, tuple(targets.shape),
                                             prob._version) and targets.dtype == torch.int64:
        return cached[4]            # CrossEntropyLoss2d counted them while it read these very tensors
    lib = _lib.lib()
    n, k, h, w = prob.shape
    scores = prob.detach()
    if scores.dtype != torch.float32 or not scores.is_contiguous():
        scores = scores.float().contiguous()
    tgt = targets.detach().reshape(n, h, w).to(torch.int64).contiguous()
    counts = torch.zeros(2, k, dtype=torch.int64, device=prob.device)
    ws = torch.empty(2048 * 2 * k * 4 + 256, dtype=torch.uint8, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(lib.erf_accuracy_counts(scores.data_ptr(), tgt.data_ptr(), n, k, h, w, counts.data_ptr(),
                                           ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream(prob.device).cuda_stream))
    return counts


def _on_gpu(prob, targets):
    return prob.is_cuda and prob.dim() == 4 and targets.is_cuda and targets.numel() == prob.numel() // prob.shape[1] \
        and 1 <= prob.shape[1] <= 256


class ClassificationAccuracy():
    """Fraction of the labelled pixels (label >= 0) whose arg-max class equals the label
    (train/evaluation_functions.py:133-139)."""

    def __call__(self, prob, targets):
        if _on_gpu(prob, targets):
            counts = _accuracy_counts(prob, targets)
            return counts[0].sum().float() / counts[1].sum().float()
        max_class = prob.argmax(dim=1, keepdim=True).squeeze()
        n_correct = (max_class == targets).sum().float()
        n_total = (torch.ge(targets, 0.0)).sum().float()
        return n_correct / n_total


class MeanAccuracy():
    """Mean over the classes present in the labels of correct / labelled pixels, and the per-class
    values (nan for absent classes) — train/evaluation_functions.py:141-163."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def __call__(self, prob, targets):
        if _on_gpu(prob, targets) and self.n_classes <= prob.shape[1]:
            counts = _accuracy_counts(prob, targets)[:, :self.n_classes].to("cpu", torch.float32)
            class_acc = counts[0] / counts[1]
            present = counts[1] > 0
            return class_acc[present].sum() / present.sum(), class_acc.numpy()
        # CPU tensors: two histograms over the label classes instead of a loop of masked counts
        pred = prob.argmax(dim=1).reshape(-1)
        label = targets.reshape(-1).to(torch.int64)
        keep = (label >= 0) & (label < self.n_classes)
        total = torch.bincount(label[keep], minlength=self.n_classes).float()
        hit = torch.bincount(label[keep & (pred == label)], minlength=self.n_classes).float()
        class_acc = hit / total
        present = total > 0
        return class_acc[present].sum() / present.sum(), class_acc.numpy()
