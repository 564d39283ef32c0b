"""Drop-in for the reference's ``train/net_classification_ladder.py`` on the HIP blocks."""
import torch
import torch.nn as nn

try:
    from ._nets import TrainDecoder, TrainEncoder, class_power_init
    from .erfnet_blocks import *  # noqa: F401,F403
except ImportError:
    from erfnet_pytorch_amd._nets import TrainDecoder, TrainEncoder, class_power_init
    from erfnet_pytorch_amd.erfnet_blocks import *  # noqa: F401,F403


class Encoder(TrainEncoder):
    def __init__(self):
        super().__init__(use_dropout=False, ladder=True)


class Decoder(TrainDecoder):
    def __init__(self, softmax_classes, late_dropout_prob):
        super().__init__(head=softmax_classes, late_dropout_prob=late_dropout_prob, ladder=True)


class Net(nn.Module):
    def __init__(self, encoder=None, softmax_classes=0, spread_class_power=False,
                 fix_class_power=False, late_dropout_prob=0.3):
        super().__init__()
        self.softmax_classes = softmax_classes
        if softmax_classes > 0:
            init = class_power_init(softmax_classes, spread_class_power, 0.7, 2.0)
            if fix_class_power:
                self.class_power = init.cuda()
            else:
                self.class_power = nn.Parameter(init)
        self.encoder = Encoder() if encoder is None else encoder
        self.decoder = Decoder(softmax_classes, late_dropout_prob)

    def forward(self, input):
        out = self.decoder(*self.encoder(input))
        if self.softmax_classes > 0:
            return out, self.class_power
        return out
