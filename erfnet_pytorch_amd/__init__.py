"""erfnet_pytorch_amd — the ERFNet block forward/backward path on MI355X (gfx950).

Sub-modules carry the reference's file names so that its scripts can import them unchanged:
``erfnet_blocks``, ``net_classification``, ``net_regression``(``_small`` / ``_ladder`` /
``_ladder_small``), ``net_classification_ladder``, ``erfnet`` (eval, BatchNorm on), ``erfnet_nobn``,
``erfnet_imagenet``.  ``functional`` holds the autograd functions, ``_lib`` the ctypes binding of
``liberfnet_hip.so`` and ``ddp`` the one-process-per-GPU data-parallel wrapper.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def build(force=False, verbose=False):
    """Compile the HIP library in-tree (hipcc, gfx950)."""
    return _lib.build(force=force, verbose=verbose)
