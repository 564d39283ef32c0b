"""Drop-in for the reference's ``iouEval.py`` (train/iouEval.py, eval/iouEval.py): same class,
constructor, ``reset`` / ``addBatch`` / ``getIoU`` and the console colour helpers.

For label maps on a GPU (the ``[N,1,H,W]`` int64 tensors the training and evaluation scripts pass)
``addBatch`` is one counting pass in ``liberfnet_hip.so`` with integer counters that stay on the
device — no one-hot tensors, no per-batch device-to-host copy.  One-hot / score inputs and CPU
tensors follow the reference's formulation with torch ops.
"""
import torch

try:
    from . import _lib
except ImportError:  # imported as a top-level module next to the reference's scripts
    from erfnet_pytorch_amd import _lib

__all__ = ["iouEval", "colors", "getColorEntry"]


class iouEval:
    def __init__(self, nClasses, ignoreIndex=19):
        self.nClasses = nClasses
        # an ignore index beyond the classes means "nothing is ignored" (train/iouEval.py:12)
        self.ignoreIndex = ignoreIndex if nClasses > ignoreIndex else -1
        self.reset()

    def reset(self):
        classes = self.nClasses if self.ignoreIndex == -1 else self.nClasses - 1
        self._host = torch.zeros(3, classes, dtype=torch.float64)      # tp, fp, fn as the reference keeps them
        self._dev = {}                                                 # device -> int64 [3][counted]

    # the reference exposes tp / fp / fn as CPU double tensors
    def _total(self):
        tot = self._host.clone()
        for buf in self._dev.values():
            c = buf.to("cpu", torch.float64)
            if c.shape[1] != tot.shape[1]:
                raise RuntimeError(f"iouEval: {c.shape[1]} counted classes do not fit {tot.shape[1]} counters "
                                   "(ignoreIndex must be the last class)")
            tot += c
        return tot

    tp = property(lambda self: self._total()[0])
    fp = property(lambda self: self._total()[1])
    fn = property(lambda self: self._total()[2])

    def addBatch(self, x, y):  # x = predictions, y = targets
        if x.is_cuda or y.is_cuda:
            dev = x.device if x.is_cuda else y.device
            x, y = x.to(dev), y.to(dev)
            if x.size(1) == 1 and y.size(1) == 1 and not x.is_floating_point() and not y.is_floating_point():
                return self._add_labels_gpu(x, y)
        self._add_torch(x, y)

    def _add_labels_gpu(self, x, y):
        lib = _lib.lib()
        x = x.to(torch.int64).contiguous()
        y = y.to(torch.int64).contiguous()
        if x.numel() != y.numel():
            raise RuntimeError("iouEval: prediction and target differ in size")
        counted = self.nClasses if self.ignoreIndex == -1 else self.ignoreIndex
        buf = self._dev.get(x.device)
        if buf is None:
            buf = self._dev[x.device] = torch.zeros(3, counted, dtype=torch.int64, device=x.device)
        ws = torch.empty(1024 * 3 * max(counted, 1) * 4 + 256, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.erf_iou_add_batch(x.data_ptr(), y.data_ptr(), x.numel(), int(self.nClasses),
                                             int(self.ignoreIndex), buf.data_ptr(), ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream(x.device).cuda_stream))

    def _add_torch(self, x, y):
        """The reference's one-hot formulation (train/iouEval.py:31-70) for everything else."""
        def onehot(t):
            if t.size(1) != 1:
                return t.float()
            z = torch.zeros(t.size(0), self.nClasses, t.size(2), t.size(3), device=t.device)
            return z.scatter_(1, t.long(), 1)
        xo, yo = onehot(x), onehot(y)
        ignores = 0
        if self.ignoreIndex != -1:
            ignores = yo[:, self.ignoreIndex].unsqueeze(1)
            xo, yo = xo[:, :self.ignoreIndex], yo[:, :self.ignoreIndex]
        dims = (0, 2, 3)
        tp = (xo * yo).sum(dims)
        fp = (xo * (1 - yo - ignores)).sum(dims)
        fn = ((1 - xo) * yo).sum(dims)
        self._host[0] += tp.double().cpu()
        self._host[1] += fp.double().cpu()
        self._host[2] += fn.double().cpu()

    def getIoU(self):
        tp, fp, fn = self._total()
        iou = tp / (tp + fp + fn + 1e-15)
        return torch.mean(iou), iou  # mean IoU, IoU per class


class colors:
    """ANSI escape sequences used by the evaluation scripts' tables."""
    RED, GREEN, YELLOW, BLUE = "\033[31;1m", "\033[32;1m", "\033[33;1m", "\033[34;1m"
    MAGENTA, CYAN = "\033[35;1m", "\033[36;1m"
    BOLD, UNDERLINE, ENDC = "\033[1m", "\033[4m", "\033[0m"


def getColorEntry(val):
    """Colour for an IoU value: red below 0.2, then yellow, blue, cyan, green from 0.8 up."""
    if not isinstance(val, float):
        return colors.ENDC
    for limit, colour in ((0.20, colors.RED), (0.40, colors.YELLOW), (0.60, colors.BLUE), (0.80, colors.CYAN)):
        if val < limit:
            return colour
    return colors.GREEN
