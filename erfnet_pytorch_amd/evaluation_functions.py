"""Drop-in for the classification loss and accuracy metrics of ``train/evaluation_functions.py``.

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


def _accuracy_counts(prob, targets):
    """int64 [2][K] on the device: per label class, correctly classified pixels and labelled pixels
    (one pass of erf_accuracy_counts over the scores, no argmax tensor, no per-class loop)."""
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
        max_class = prob.argmax(dim=1, keepdim=True).squeeze()
        correct_pred = max_class.masked_select(max_class == targets)
        mean_acc, num_classes, class_acc = 0, 0, []
        for cur_class in range(0, self.n_classes):
            n_correct = (correct_pred == cur_class).sum().float()
            n_total = (targets == cur_class).sum().float()
            cur_acc = n_correct / n_total
            class_acc.append(float(cur_acc))
            if n_total > 0:
                num_classes += 1
                mean_acc += cur_acc
        return mean_acc / num_classes, np.array(class_acc)
