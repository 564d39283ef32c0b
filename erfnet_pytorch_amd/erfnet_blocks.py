"""ERFNet building blocks with the module API of the reference's ``train/erfnet_blocks.py``
(same class names, constructor arguments, sub-module names and therefore ``state_dict`` keys),
computing on MI355X through the HIP kernels of ``liberfnet_hip.so``.

Dispatch rule of every block's ``forward``:
  * tensor on a GPU  -> the HIP kernels (``functional.py``); a missing library raises, nothing else
    is tried;
  * tensor on the CPU -> the block is evaluated with stock ``torch.nn.functional`` ops on the same
    parameters.  This is host-side plumbing only (state_dict round trips, CPU smoke runs such as
    ``eval_forwardTime.py --cpu``); it is never reached for GPU tensors.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from . import functional as hipf
except ImportError:  # imported as a top-level module next to the reference's scripts
    from erfnet_pytorch_amd import functional as hipf

__all__ = ["DownsamplerBlock", "non_bottleneck_1d", "UpsamplerBlock", "SoftMaxConv",
           "DecoderBlockSmall", "DecoderBlock", "LadderBlock", "run_layers"]

_BN_EPS = 1e-3


def _bn_mode(module, enabled):
    """0 = BatchNorm skipped, 1 = running statistics, 2 = batch statistics."""
    if not enabled:
        return 0
    return 2 if module.training else 1


def _bn_args(bn):
    """The tensors of a BatchNorm2d for the library.  num_batches_tracked travels as an attribute of
    running_mean: the library's statistics kernel adds 1 to it in a train-mode forward, as
    nn.BatchNorm2d.forward does (no separate launch per layer)."""
    rm, nbt = bn.running_mean, bn.num_batches_tracked
    if rm is not None:
        rm._erf_nbt = nbt if (nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64) else None
    return bn.weight, bn.bias, rm, bn.running_var


def _bn_tick(bn, mode):
    """Counter update for the cases the library did not cover (counter not an int64 GPU tensor)."""
    if mode == 2 and bn.num_batches_tracked is not None and getattr(bn.running_mean, "_erf_nbt", None) is None:
        bn.num_batches_tracked += 1


def _dropout2d_scale(x, p, active):
    """Per-(n,c) keep/scale factors of nn.Dropout2d, or None when dropout does nothing."""
    if not active or p == 0:
        return None
    n, c = x.shape[0], x.shape[1]
    if p >= 1:
        return torch.zeros(n, c, dtype=torch.float32, device=x.device)
    return torch.empty(n, c, dtype=torch.float32, device=x.device).bernoulli_(1.0 - p).div_(1.0 - p)


class DownsamplerBlock(nn.Module):
    """relu([bn](cat(conv3x3/2(x), maxpool2x2(x)))) — reference train/erfnet_blocks.py:15-29."""

    def __init__(self, ninput, noutput, use_batchnorm=False):
        super().__init__()
        self.use_batchnorm = use_batchnorm
        self.conv = nn.Conv2d(ninput, noutput - ninput, (3, 3), stride=2, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2)
        self.bn = nn.BatchNorm2d(noutput, eps=_BN_EPS)

    def forward(self, input):
        if input.is_cuda:
            mode = _bn_mode(self, self.use_batchnorm)
            out = hipf.downsampler(input, self.conv.weight, self.conv.bias, *_bn_args(self.bn),
                                   bn_mode=mode, momentum=self.bn.momentum, eps=self.bn.eps)
            _bn_tick(self.bn, mode)
            return out
        both = torch.cat((self.conv(input), self.pool(input)), dim=1)
        return F.relu(self.bn(both) if self.use_batchnorm else both)


class non_bottleneck_1d(nn.Module):
    """Factorised residual block — reference train/erfnet_blocks.py:33-73."""

    _CONVS = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")

    # Registration order of the sub-modules = order of the state_dict keys.  The train/ and eval/
    # copies interleave (conv, conv, bn1, conv, conv, bn2); imagenet/erfnet_imagenet.py:29-36
    # registers the four convolutions first.
    _ORDER = ("conv3x1_1", "conv1x3_1", "bn1", "conv3x1_2", "conv1x3_2", "bn2")

    def __init__(self, chann, dropprob, dilated, use_dropout=False, use_batchnorm=False):
        super().__init__()
        self.use_dropout = use_dropout
        self.use_batchnorm = use_batchnorm
        self.dilated = dilated
        d = dilated
        parts = {
            "conv3x1_1": nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1, 0), bias=True),
            "conv1x3_1": nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1), bias=True),
            "bn1": nn.BatchNorm2d(chann, eps=_BN_EPS),
            "conv3x1_2": nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(d, 0), bias=True,
                                   dilation=(d, 1)),
            "conv1x3_2": nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, d), bias=True,
                                   dilation=(1, d)),
            "bn2": nn.BatchNorm2d(chann, eps=_BN_EPS),
        }
        for name in self._ORDER:
            setattr(self, name, parts[name])
        self.dropout = nn.Dropout2d(dropprob)

    def _dropout_enabled(self):
        return self.use_dropout and self.dropout.p != 0

    def _gpu_args(self, input, scale=False):
        """Keyword arguments of functional.nb1d for this block.  ``scale``: the block's dropout factors
        when the caller has drawn them (None: dropout does nothing), False: drawn here."""
        mods = self._modules            # direct look-ups: this runs once per block per forward
        convs = [mods[n] for n in self._CONVS]
        mode = _bn_mode(self, self.use_batchnorm)
        if scale is False:
            scale = None
            if self.use_dropout and self.training:
                scale = _dropout2d_scale(input, mods["dropout"].p, self._dropout_enabled())
        kw = dict(weights=[c._parameters["weight"] for c in convs], biases=[c._parameters["bias"] for c in convs],
                  dilated=self.dilated, bn_mode=mode, drop_scale=scale)
        if mode:
            bn1, bn2 = mods["bn1"], mods["bn2"]
            kw.update(bn1=_bn_args(bn1), bn2=_bn_args(bn2), momentum=bn1.momentum, eps=bn1.eps)
        return kw

    def forward(self, input):
        if input.is_cuda:
            kw = self._gpu_args(input)
            out = hipf.nb1d(input, kw.pop("weights"), kw.pop("biases"), kw.pop("dilated"), **kw)
            _bn_tick(self.bn1, kw["bn_mode"])
            _bn_tick(self.bn2, kw["bn_mode"])
            return out
        t = F.relu(self.conv3x1_1(input))
        t = self.conv1x3_1(t)
        t = F.relu(self.bn1(t) if self.use_batchnorm else t)
        t = F.relu(self.conv3x1_2(t))
        t = self.conv1x3_2(t)
        if self.use_batchnorm:
            t = self.bn2(t)
        if self._dropout_enabled():
            t = self.dropout(t)
        return F.relu(t + input)


def _plain_call(m):
    """True when calling m.forward directly equals m(...): no hooks are registered."""
    return not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def _run_dropout_scales(run, x):
    """Dropout2d factors of a run of blocks: one draw for the whole run when every block drops with the
    same probability (two launches per run instead of two per block), else block by block (False)."""
    ps = {m.dropout.p if (m.use_dropout and m.training and m._dropout_enabled()) else 0.0 for m in run}
    if len(ps) != 1:
        return [False] * len(run)
    p = ps.pop()
    if p == 0:
        return [None] * len(run)
    if p >= 1:
        return [False] * len(run)
    n, c = x.shape[0], x.shape[1]
    both = torch.empty(len(run), n, c, dtype=torch.float32, device=x.device).bernoulli_(1.0 - p).div_(1.0 - p)
    return list(both.unbind(0))


def run_layers(layers, x):
    """``for layer in layers: x = layer(x)`` (reference train/net_classification.py:32-36, 42-45).
    On a GPU, runs of consecutive non_bottleneck_1d blocks of one width go to the library as one
    chain (functional.nb1d_chain), which lets it schedule the last convolution of a block and the
    first of the next as a single launch; the values are the same as block-by-block calls."""
    layers = list(layers)
    i = 0
    while i < len(layers):
        j = i
        if x.is_cuda:
            while (j < len(layers) and isinstance(layers[j], non_bottleneck_1d) and _plain_call(layers[j])
                   and type(layers[j]).forward is non_bottleneck_1d.forward
                   and layers[j].conv3x1_1.in_channels == x.shape[1]):
                j += 1
        if j - i >= 2:
            run = layers[i:j]
            x = hipf.nb1d_chain(x, [m._gpu_args(x, sc) for m, sc in zip(run, _run_dropout_scales(run, x))])
            for m in run:
                mode = _bn_mode(m, m.use_batchnorm)
                _bn_tick(m.bn1, mode)
                _bn_tick(m.bn2, mode)
            i = j
        else:
            x = layers[i](x)
            i += 1
    return x


class UpsamplerBlock(nn.Module):
    """relu([bn](convT3x3/2(x))) — reference train/erfnet_blocks.py:77-88."""

    def __init__(self, ninput, noutput, use_batchnorm=False):
        super().__init__()
        self.use_batchnorm = use_batchnorm
        self.conv = nn.ConvTranspose2d(ninput, noutput, 3, stride=2, padding=1, output_padding=1,
                                       bias=True)
        self.bn = nn.BatchNorm2d(noutput, eps=_BN_EPS)

    def forward(self, input):
        if input.is_cuda:
            mode = _bn_mode(self, self.use_batchnorm)
            out = hipf.upsampler(input, self.conv.weight, self.conv.bias, *_bn_args(self.bn),
                                 bn_mode=mode, momentum=self.bn.momentum, eps=self.bn.eps)
            _bn_tick(self.bn, mode)
            return out
        t = self.conv(input)
        return F.relu(self.bn(t) if self.use_batchnorm else t)


class OutputConvTranspose(nn.ConvTranspose2d):
    """``nn.ConvTranspose2d(cin, classes, 2, stride=2)`` (the networks' last layer; reference
    train/net_classification.py:56) whose GPU forward/backward run in the HIP kernels.  It is an
    ``nn.ConvTranspose2d`` in every other respect (parameters, state_dict, repr)."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 2, stride=2, padding=0, output_padding=0,
                         bias=True)

    def forward(self, input, output_size=None):
        if input.is_cuda:
            # (a DataParallel replica's output is gathered in C++, which a deferred tensor cannot notice)
            return hipf.outconv(input, self.weight, self.bias, allow_lazy=not getattr(self, "_is_replica", False))
        return super().forward(input, output_size)

    def argmax(self, input):
        """``self(input).max(1)[1].unsqueeze(1)`` (what eval/eval_iou.py:92 feeds to iouEval); on a
        GPU one kernel that never writes the logits."""
        if input.is_cuda:
            return hipf.outconv_argmax(input, self.weight, self.bias)
        return super().forward(input).max(1)[1].unsqueeze(1)


class SoftMaxConv(nn.Module):
    """convT2x2/2 -> [Dropout2d] -> softmax(dim=1) — reference train/erfnet_blocks.py:92-110."""

    def __init__(self, in_channels, softmax_classes, late_dropout_prob, use_dropout):
        super().__init__()
        self.use_dropout = use_dropout
        self.convolution = OutputConvTranspose(in_channels, softmax_classes)
        self.dropout = torch.nn.Dropout2d(p=late_dropout_prob)

    def forward(self, input):
        if input.is_cuda and not torch.is_grad_enabled() and not (self.use_dropout and self.training):
            return hipf.outconv_softmax(input, self.convolution.weight, self.convolution.bias)
        t = self.convolution(input)
        if self.use_dropout:
            t = self.dropout(t)
        return F.softmax(t, dim=1)


class DecoderBlockSmall(nn.Module):
    """Upsampler + one non_bottleneck_1d — reference train/erfnet_blocks.py:114-129."""

    def __init__(self, in_channels, out_channels, use_dropout=False):
        super().__init__()
        self.layers = nn.ModuleList([
            UpsamplerBlock(in_channels, out_channels),
            non_bottleneck_1d(out_channels, 0, 1, use_dropout=use_dropout),
        ])

    def forward(self, input):
        return run_layers(self.layers, input)


class DecoderBlock(DecoderBlockSmall):
    """Upsampler + two non_bottleneck_1d — reference train/erfnet_blocks.py:133-136."""

    def __init__(self, in_channels, out_channels, use_dropout=False):
        super().__init__(in_channels, out_channels, use_dropout)
        self.layers.append(non_bottleneck_1d(out_channels, 0, 1, use_dropout=use_dropout))


class LadderBlock(nn.Module):
    """relu([bn](conv3x3(nb1d(cat(x, skip))))) — reference train/erfnet_blocks.py:140-154."""

    def __init__(self, noutput, use_batchnorm=False, use_dropout=False):
        super().__init__()
        self.use_batchnorm = use_batchnorm
        self.layer = non_bottleneck_1d(2 * noutput, 0.03, 1, use_dropout=use_dropout)
        self.conv = nn.Conv2d(2 * noutput, noutput, 3, stride=1, padding=1, bias=True)
        self.bn = nn.BatchNorm2d(noutput, eps=_BN_EPS)

    def forward(self, input, input_ladder):
        t = self.layer(torch.cat((input, input_ladder), dim=1))
        if t.is_cuda and not self.use_batchnorm:
            return hipf.conv2d(t, self.conv.weight, self.conv.bias, relu=True)
        if t.is_cuda:
            t = hipf.conv2d(t, self.conv.weight, self.conv.bias, relu=False)
            return F.relu(self.bn(t))
        t = self.conv(t)
        return F.relu(self.bn(t) if self.use_batchnorm else t)
