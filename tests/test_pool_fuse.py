"""DownsamplerBlock(16, 64) backward with the max-pool gradient formed in the data-gradient kernel's
epilogue (ERF_OPT_POOL_FUSE) against the separate pool kernel: every gradient bit-identical, one launch
fewer.  Reference block: train/erfnet_blocks.py:15-29."""
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(fused, dtype, relu_input):
    from erfnet_pytorch_amd import _lib, erfnet_blocks
    prev = _lib.set_option(_lib.OPT_POOL_FUSE, int(fused))
    try:
        block = gu.fill_state_dict(erfnet_blocks.DownsamplerBlock(16, 64), tag="poolfuse").to(DEV)
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn(2, 16, 36, 52, generator=g)
        if relu_input:                      # ties inside a pooling window (zeros): the first one takes the gradient
            x = x.clamp_min(0)
        x = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gy = torch.randn(2, 64, 18, 26, generator=g).to(DEV).to(dtype)
        y = block(x)
        before = _lib.launch_count()
        y.backward(gy)
        torch.cuda.synchronize()
        n = _lib.launch_count() - before
        return [x.grad] + [p.grad for p in block.parameters()], n
    finally:
        _lib.set_option(_lib.OPT_POOL_FUSE, prev)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu_input", [False, True])
def test_pool_gradient_in_the_epilogue_is_bit_identical(dtype, relu_input):
    ref, n_ref = _run(False, dtype, relu_input)
    got, n_got = _run(True, dtype, relu_input)
    for a, b in zip(ref, got):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert n_got == n_ref - 1, (n_got, n_ref)
