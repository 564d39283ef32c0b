"""What the gradient all-reduces cost the backward they overlap with, measured on ONE GPU: bench.py's
training step with the reducer's bucket collectives really launched over RCCL (a one-rank group:
ncclAllReduce degenerates to a device copy kernel on RCCL's stream, but it is enqueued, ordered against
the compute stream and competes for CUs with the persistent one-block-per-CU convolution kernels exactly
where the multi-rank job launches its rings) against the same step with the collectives skipped:
`ddp_overlap_probe.py [f32|bf16] [steps]`."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import net_classification, optim  # noqa: E402
from erfnet_pytorch_amd.ddp import GradientReducer  # noqa: E402
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d  # noqa: E402

DEV = "cuda:0"
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)


def rate(always):
    torch.manual_seed(0)
    model = net_classification.Net(softmax_classes=20).to(DEV).train()
    red = GradientReducer(model, bucket_bytes=3_600_000, direct_write=True, reduce_single_rank=always)
    crit = CrossEntropyLoss2d()
    x = torch.rand(6, 3, 512, 1024, device=DEV).to(dtype)
    lab = torch.randint(-1, 20, (6, 512, 1024), device=DEV)
    red.zero_grad()
    crit(model(x)[0], lab).backward()
    red.finish()
    opt = optim.Adam(red.flatten_parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)

    def step():
        red.zero_grad()
        loss = crit(model(x)[0], lab)
        loss.backward()
        red.finish()
        opt.step()
        return loss

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, float(loss)


for rnd in range(2):
    for always in (False, True):
        dt, loss = rate(always)
        print(f"collectives {'launched' if always else 'skipped '}: {dt * 1e3:.3f} ms/step  {6 / dt:.1f} img/s  loss {loss:.5f}", flush=True)
dist.destroy_process_group()
