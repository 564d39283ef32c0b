"""The RCCL path of the gradient reducer on one GPU: a one-rank nccl group with the collectives
forced on (bucketed in-place all-reduce enqueued from inside backward, flat parameter broadcast,
direct gradient write) must leave exactly the gradients and parameters of a plain training loop.
Two ranks cannot share a GPU under RCCL; the rank-to-rank arithmetic is covered by the world-2
gloo test on CPU (tests/test_ddp_cpu.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu
from test_boundary_cpu import make_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(reducer_kwargs, steps=3):
    from erfnet_pytorch_amd.ddp import GradientReducer
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    net = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV).train()
    x = torch.as_tensor(gu.det_uniform("trainstep.x", (2, 3, 64, 128), 0.0, 1.0)).to(DEV)
    lab = torch.as_tensor(gu.det_labels("trainstep.lab", (2, 64, 128), -1, 20)).to(DEV)
    red = GradientReducer(net, **reducer_kwargs) if reducer_kwargs is not None else None
    if red is not None:
        red.broadcast_parameters(0)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], 1e-4, eps=1e-4)
    crit = CrossEntropyLoss2d()
    grads = []
    for _ in range(steps):
        red.zero_grad() if red is not None else opt.zero_grad()
        out, _ = net(x)
        crit(out, lab).backward()
        if red is not None:
            red.finish()
        grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        opt.step()
    torch.cuda.synchronize()
    if red is not None:
        red.remove()
    return grads, {k: p.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.timeout(240)
def test_one_rank_nccl_group_with_forced_collectives():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        g_ref, p_ref = _run(None)
        g_red, p_red = _run(dict(direct_write=True, reduce_single_rank=True, bucket_bytes=1 << 20))
    finally:
        dist.destroy_process_group()
    for step, (a, b) in enumerate(zip(g_ref, g_red)):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (step, k)
    for k in p_ref:
        assert torch.equal(p_ref[k], p_red[k]), k
