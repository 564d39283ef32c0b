"""Drop-in for the reference's ``imagenet/erfnet_imagenet.py``: the ERFNet encoder (BatchNorm on,
dropout 0.1) followed by max-pool, 14x14 average pool and a linear classifier.  The encoder blocks
run in the HIP kernels; the tiny pooling / linear head stays stock torch."""
import torch.nn as nn

try:
    from . import _nets
except ImportError:
    from erfnet_pytorch_amd import _nets

DownsamplerBlock = _nets.BnDownsamplerBlock
non_bottleneck_1d = _nets.BnNonBottleneck1dConvsFirst


class Encoder(nn.Module):
    """reference imagenet/erfnet_imagenet.py:60-86 (no output_conv in this variant)"""

    def __init__(self):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16)
        self.layers = nn.ModuleList(_nets.eval_encoder_layers(DownsamplerBlock, non_bottleneck_1d, 0.1))

    def forward(self, input):
        t = self.initial_block(input)
        t = _nets.run_layers(self.layers, t)
        return t


class Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.extralayer1 = nn.MaxPool2d(2, stride=2)
        self.extralayer2 = nn.AvgPool2d(14, 1, 0)

    def forward(self, input):
        return self.extralayer2(self.extralayer1(self.encoder(input)))


class Classifier(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.linear = nn.Linear(128, num_classes)

    def forward(self, input):
        t = input.reshape(input.size(0), 128)
        # bf16 activations with fp32 master parameters: the 128 -> classes product runs in the
        # parameters' type
        return self.linear(t.to(self.linear.weight.dtype))


class ERFNet(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.features = Features()
        self.classifier = Classifier(num_classes)

    def forward(self, input):
        return self.classifier(self.features(input))
