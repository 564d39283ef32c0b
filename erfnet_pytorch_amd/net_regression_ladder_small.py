"""Drop-in for the reference's ``train/net_regression_ladder_small.py`` (``Encoder`` / ``Decoder`` / ``Net`` with
identical constructor arguments, attribute names and state_dict keys) on the HIP blocks."""
import torch
import torch.nn as nn

try:
    from ._nets import TrainDecoder, TrainEncoder, class_power_init
    from .erfnet_blocks import *  # noqa: F401,F403  (the reference re-exports the blocks)
except ImportError:
    from erfnet_pytorch_amd._nets import TrainDecoder, TrainEncoder, class_power_init
    from erfnet_pytorch_amd.erfnet_blocks import *  # noqa: F401,F403

_SMALL, _LADDER, _LATE_DROPOUT = True, True, 0.0


class Encoder(TrainEncoder):
    def __init__(self, use_dropout=False):
        super().__init__(use_dropout=use_dropout, n64=2 if _SMALL else 5,
                         reps128=1 if _SMALL else 2, ladder=_LADDER)


class Decoder(TrainDecoder):
    def __init__(self, softmax_classes, late_dropout_prob, likelihood_loss, use_dropout=False):
        if softmax_classes:
            head = "softmax"
        else:
            head = 2 if likelihood_loss else 1   # (mean, variance) or a single scalar map
        super().__init__(head=head, late_dropout_prob=late_dropout_prob, use_dropout=use_dropout,
                         small=_SMALL, ladder=_LADDER, softmax_classes=softmax_classes)


class Net(nn.Module):
    """reference train/net_regression_ladder_small.py: Net"""

    def __init__(self, encoder=None, softmax_classes=0, likelihood_loss=False,
                 spread_class_power=False, late_dropout_prob=_LATE_DROPOUT):
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
        out = self.decoder(*enc) if _LADDER else self.decoder(enc)
        if self.softmax_classes > 0:
            if self.likelihood_loss:
                return out, self.class_power, self.class_power_var
            return out, self.class_power
        return out
