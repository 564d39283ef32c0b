"""Drop-in for the reference's ``train/net_regression.py`` (``Encoder`` / ``Decoder`` / ``Net`` with
identical constructor arguments, attribute names and state_dict keys) on the HIP blocks."""
try:
    from ._nets import regression_classes
    from .erfnet_blocks import *  # noqa: F401,F403  (the reference re-exports the blocks)
except ImportError:
    from erfnet_pytorch_amd._nets import regression_classes
    from erfnet_pytorch_amd.erfnet_blocks import *  # noqa: F401,F403

Encoder, Decoder, Net = regression_classes(__name__, small=False, ladder=False)
