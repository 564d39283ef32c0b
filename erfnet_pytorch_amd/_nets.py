"""Network assembly shared by the drop-in model files (``net_*.py``, ``erfnet*.py``).

Only composition lives here: which block follows which, under the attribute names the reference
uses (they are the ``state_dict`` keys: reference train/net_classification.py:17-66,
train/net_regression_ladder.py:16-80, eval/erfnet.py:64-152, imagenet/erfnet_imagenet.py:60-122).
"""
import numpy as np
import torch
import torch.nn as nn

from . import functional as hipf
from .erfnet_blocks import (DecoderBlock, DecoderBlockSmall, DownsamplerBlock, LadderBlock,
                            OutputConvTranspose, SoftMaxConv, UpsamplerBlock, non_bottleneck_1d,
                            run_layers)

DILATIONS = (2, 4, 8, 16)


def conv1x1(module, x):
    """Encoder.output_conv on either device."""
    if x.is_cuda:
        return hipf.conv2d(x, module.weight, module.bias, relu=False)
    return module(x)


def stage64(make_nb, count, p):
    return [make_nb(64, p, 1) for _ in range(count)]


def stage128(make_nb, repeats, p):
    return [make_nb(128, p, d) for _ in range(repeats) for d in DILATIONS]


class TrainEncoder(nn.Module):
    """Encoder of the train/ networks: BN off, dropout only when ``use_dropout``.

    ladder=False registers one ``layers`` list; ladder=True registers ``block1`` / ``block2`` and
    returns the three resolutions (reference train/net_regression_ladder.py:16-52)."""

    def __init__(self, use_dropout=False, n64=5, reps128=2, ladder=False):
        super().__init__()
        self.ladder = ladder

        def nb(c, p, d):
            return non_bottleneck_1d(c, p, d, use_dropout=use_dropout)

        self.initial_block = DownsamplerBlock(3, 16)
        first = [DownsamplerBlock(16, 64)] + stage64(nb, n64, 0.03)
        second = [DownsamplerBlock(64, 128)] + stage128(nb, reps128, 0.3)
        if ladder:
            self.block1 = nn.ModuleList(first)
            self.block2 = nn.ModuleList(second)
        else:
            self.layers = nn.ModuleList(first + second)
        # only used in encoder-only mode, kept for state_dict compatibility
        self.output_conv = nn.Conv2d(128, 1, 1, stride=1, padding=0, bias=True)

    def forward(self, input):
        out1 = self.initial_block(input)
        if not self.ladder:
            return run_layers(self.layers, out1)
        out2 = run_layers(self.block1, out1)
        out3 = run_layers(self.block2, out2)
        return out1, out2, out3


class TrainDecoder(nn.Module):
    """Decoder of the train/ networks (reference train/net_regression.py:48-72 and the ladder /
    small variants).  head: 'softmax' -> SoftMaxConv, or an int -> plain ConvTranspose2d(16, head)."""

    def __init__(self, head, late_dropout_prob=0.0, use_dropout=False, small=False, ladder=False,
                 softmax_classes=0):
        super().__init__()
        block = DecoderBlockSmall if small else DecoderBlock
        self.ladder = ladder
        self.scalar_decoder_1 = block(128, 64, use_dropout=use_dropout)
        if ladder:
            self.ladder_block_1 = LadderBlock(64, use_dropout=use_dropout)
        self.scalar_decoder_2 = block(64, 16, use_dropout=use_dropout)
        if ladder:
            self.ladder_block_2 = LadderBlock(16, use_dropout=use_dropout)
        if head == "softmax":
            self.scalar_output_conv = SoftMaxConv(16, softmax_classes, late_dropout_prob,
                                                  use_dropout=use_dropout)
        else:
            self.scalar_output_conv = OutputConvTranspose(16, head)

    def forward(self, *enc):
        if self.ladder:
            enc1, enc2, enc3 = enc
            t = self.scalar_decoder_1(enc3)
            t = self.ladder_block_1(t, enc2)
            t = self.scalar_decoder_2(t)
            t = self.ladder_block_2(t, enc1)
        else:
            (t,) = enc
            t = self.scalar_decoder_2(self.scalar_decoder_1(t))
        return self.scalar_output_conv(t)


def class_power_init(softmax_classes, spread, lo, hi):
    if spread:
        vals = np.linspace(lo, hi, softmax_classes).astype("float32").reshape(1, softmax_classes, 1, 1)
        return torch.from_numpy(vals)
    return torch.ones(1, softmax_classes, 1, 1)


# ------------------------------------------------------------------------- eval/ and imagenet/ family
class BnDownsamplerBlock(DownsamplerBlock):
    def __init__(self, ninput, noutput):
        super().__init__(ninput, noutput, use_batchnorm=True)


class BnUpsamplerBlock(UpsamplerBlock):
    def __init__(self, ninput, noutput):
        super().__init__(ninput, noutput, use_batchnorm=True)


class BnNonBottleneck1d(non_bottleneck_1d):
    """BN always on; Dropout2d whenever its probability is non-zero (eval/erfnet.py:26-61)."""

    def __init__(self, chann, dropprob, dilated):
        super().__init__(chann, dropprob, dilated, use_dropout=True, use_batchnorm=True)


class BnNonBottleneck1dConvsFirst(BnNonBottleneck1d):
    """Same block, sub-modules registered as in imagenet/erfnet_imagenet.py:29-36."""
    _ORDER = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2", "bn1", "bn2")


class PlainDownsamplerBlock(DownsamplerBlock):
    def __init__(self, ninput, noutput):
        super().__init__(ninput, noutput, use_batchnorm=False)


class PlainUpsamplerBlock(UpsamplerBlock):
    def __init__(self, ninput, noutput):
        super().__init__(ninput, noutput, use_batchnorm=False)


class PlainNonBottleneck1d(non_bottleneck_1d):
    """BN and dropout modules exist (state_dict) but are never applied (eval/erfnet_nobn.py:27-63)."""

    def __init__(self, chann, dropprob, dilated):
        super().__init__(chann, dropprob, dilated, use_dropout=False, use_batchnorm=False)


def eval_encoder_layers(down, nb, p):
    return [down(16, 64)] + stage64(nb, 5, p) + [down(64, 128)] + stage128(nb, 2, p)


def eval_decoder_layers(up, nb):
    return [up(128, 64), nb(64, 0, 1), nb(64, 0, 1), up(64, 16), nb(16, 0, 1), nb(16, 0, 1)]


def regression_classes(module_name, small, ladder, late_dropout=0.1):
    """``Encoder`` / ``Decoder`` / ``Net`` of one of the reference's four regression network files
    (train/net_regression.py, net_regression_small.py, net_regression_ladder.py,
    net_regression_ladder_small.py: the same three classes, differing in the number of 64- and
    128-channel blocks and in the ladder connections), bound to ``module_name`` so that they pickle and
    print as that file's classes."""
    ref = "train/" + module_name.rsplit(".", 1)[-1] + ".py"

    class Encoder(TrainEncoder):
        def __init__(self, use_dropout=False):
            super().__init__(use_dropout=use_dropout, n64=2 if small else 5,
                             reps128=1 if small else 2, ladder=ladder)

    class Decoder(TrainDecoder):
        def __init__(self, softmax_classes, late_dropout_prob, likelihood_loss, use_dropout=False):
            if softmax_classes:
                head = "softmax"
            else:
                head = 2 if likelihood_loss else 1   # (mean, variance) or a single scalar map
            super().__init__(head=head, late_dropout_prob=late_dropout_prob, use_dropout=use_dropout,
                             small=small, ladder=ladder, softmax_classes=softmax_classes)

    class Net(nn.Module):
        __doc__ = f"reference {ref}: Net"

        def __init__(self, encoder=None, softmax_classes=0, likelihood_loss=False,
                     spread_class_power=False, late_dropout_prob=late_dropout):
            super().__init__()
            self.softmax_classes = softmax_classes
            self.likelihood_loss = likelihood_loss
            use_dropout = softmax_classes > 0
            if softmax_classes > 0:
                self.class_power = nn.Parameter(class_power_init(softmax_classes, spread_class_power,
                                                                 0.0, 2.0))
                self.class_power_var = nn.Parameter(torch.ones(1, softmax_classes, 1, 1))
            self.encoder = Encoder(use_dropout=use_dropout) if encoder is None else encoder
            self.decoder = Decoder(softmax_classes, late_dropout_prob, likelihood_loss,
                                   use_dropout=use_dropout)

        def forward(self, input):
            enc = self.encoder(input)
            out = self.decoder(*enc) if ladder else self.decoder(enc)
            if self.softmax_classes > 0:
                if self.likelihood_loss:
                    return out, self.class_power, self.class_power_var
                return out, self.class_power
            return out

    for cls in (Encoder, Decoder, Net):
        cls.__module__ = module_name
        cls.__qualname__ = cls.__name__
    return Encoder, Decoder, Net
