"""world_size-2 gloo test of the bucketed gradient reducer (erfnet_pytorch_amd/ddp.py) on CPU:
averaged gradients must equal the single-process gradients of the concatenated batch, unused
parameters must be dropped after the first step, and replicas must stay identical."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    from erfnet_pytorch_amd import net_regression_small
    torch.manual_seed(5)
    return net_regression_small.Net()     # 1 output channel, BatchNorm off, unused bn.* / output_conv


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.rand(2, 3, 32, 64, generator=g), torch.rand(2, 1, 32, 64, generator=g)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from erfnet_pytorch_amd.ddp import GradientReducer
    model = _model()
    if rank == 1:   # perturb: broadcast must restore rank 0's parameters
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradientReducer(model, bucket_bytes=256 << 10)
    red.broadcast_parameters(0)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
    x, y = _batch(rank)
    snaps = []
    for _ in range(3):
        red.zero_grad()
        loss = torch.nn.functional.l1_loss(model(x), y)
        loss.backward()
        red.finish()
        snaps.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        opt.step()
    torch.save({"grads": snaps, "params": {k: p.detach().clone() for k, p in model.named_parameters()},
                "buckets": red.bucket_sizes, "n_active": len(red._active)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _worker_flat(rank, world, port, out_dir):
    """flatten_parameters() + Adam on the flat buffers at world size 2, direct_write on (inert on CPU
    tensors, but the claim/deliver bookkeeping runs), a step in which the parameters of one bucket
    receive no gradient, and a step with two backward passes (deferred launches)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from erfnet_pytorch_amd.ddp import GradientReducer
    model = _model()
    red = GradientReducer(model, bucket_bytes=256 << 10, direct_write=True)
    red.broadcast_parameters(0)
    x, y = _batch(rank)
    red.zero_grad()
    torch.nn.functional.l1_loss(model(x), y).backward()
    red.finish()
    params = red.flatten_parameters()
    opt = torch.optim.Adam(params, 1e-3, eps=1e-4, weight_decay=1e-6)
    log = {}
    # 1: ordinary step
    red.zero_grad()
    torch.nn.functional.l1_loss(model(x), y).backward()
    red.finish()
    opt.step()
    # 2: only the encoder takes part (the decoder's buckets get no gradient and must still be reduced:
    # zeros on both ranks, and nothing may hang)
    red.zero_grad()
    model.encoder(x).abs().mean().backward()
    red.finish()
    log["dec_grad_zero"] = all(float(p.grad.abs().max()) == 0.0 for p in model.decoder.parameters()
                               if p.grad is not None)
    opt.step()
    # 3: two backward passes in one step
    red.zero_grad(defer=True)
    out = model(x)
    (0.5 * out.abs().mean()).backward(retain_graph=True)
    torch.nn.functional.l1_loss(out, y).backward()
    red.finish()
    log["grads3"] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    opt.step()
    # 4: the same two passes without defer must raise rather than lose a gradient
    red.zero_grad()
    out = model(x)
    (0.5 * out.abs().mean()).backward(retain_graph=True)
    try:
        torch.nn.functional.l1_loss(out, y).backward()
        log["raised"] = False
    except RuntimeError as e:
        log["raised"] = "defer=True" in str(e)
    for work, _ in red._work:        # drain what step 4 launched
        work.wait()
    red._work = []
    log["params"] = {k: p.detach().clone() for k, p in model.named_parameters()}
    torch.save(log, os.path.join(out_dir, f"flat{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_parameters_unused_bucket_and_two_backward_passes_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker_flat, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "flat0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "flat1.pt"))
    assert r0["dec_grad_zero"] and r1["dec_grad_zero"]
    assert r0["raised"] and r1["raised"]
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k          # replicas stay identical

    # single-process reference of the three updates on the concatenated batch
    model = _model()
    xs, ys = zip(*[_batch(r) for r in range(world)])
    # per-rank losses are means over the rank's batch; their average = mean over both (equal sizes)
    opt = torch.optim.Adam([p for p in model.parameters()], 1e-3, eps=1e-4, weight_decay=1e-6)
    def avg(fn):
        opt.zero_grad()
        for x, y in zip(xs, ys):
            (fn(x, y) / world).backward()
    avg(lambda x, y: torch.nn.functional.l1_loss(model(x), y))
    opt.step()
    avg(lambda x, y: model.encoder(x).abs().mean())
    # Adam skips parameters without a gradient; the reducer hands the decoder zeros instead, which
    # still moves them through the moment estimates and the weight decay: apply the same zeros
    live = set(r0["grads3"])
    for k, p in model.named_parameters():
        if k in live and p.grad is None:
            p.grad = torch.zeros_like(p)
    opt.step()
    def two(x, y):
        out = model(x)
        return 0.5 * out.abs().mean() + torch.nn.functional.l1_loss(out, y)
    avg(two)
    for k, g in r0["grads3"].items():
        torch.testing.assert_close(g, dict(model.named_parameters())[k].grad, rtol=2e-4, atol=1e-6)
    opt.step()
    for k, p in model.named_parameters():
        torch.testing.assert_close(r0["params"][k], p.detach(), rtol=2e-4, atol=2e-6)


@pytest.mark.timeout(300)
def test_reducer_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))

    # single-process reference: same model, both ranks' batches concatenated, same SGD
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    xs, ys = zip(*[_batch(r) for r in range(world)])
    x, y = torch.cat(xs), torch.cat(ys)
    for step in range(3):
        opt.zero_grad()
        torch.nn.functional.l1_loss(model(x), y).backward()
        ref = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        assert set(ref) == set(r0["grads"][step]), "set of parameters with gradients differs"
        for k, g in ref.items():
            torch.testing.assert_close(r0["grads"][step][k], g, rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(r1["grads"][step][k], r0["grads"][step][k], rtol=0, atol=0)
        opt.step()
    for k, p in model.named_parameters():
        torch.testing.assert_close(r0["params"][k], p.detach(), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(r1["params"][k], r0["params"][k], rtol=0, atol=0)
    # unused parameters (bn.*, encoder.output_conv.*) are not exchanged
    n_all = sum(1 for _ in model.parameters())
    assert r0["n_active"] == len(ref) < n_all
    assert len(r0["buckets"]) >= 2


def test_flat_parameters_give_the_same_adam_steps():
    """GradientReducer.flatten_parameters(): Adam on the flat buffers (one per gradient bucket) must
    leave every module parameter bit-identical to Adam on the individual tensors, parameters that
    never receive a gradient stay untouched, and the module keeps working on the views."""
    sys.path.insert(0, ROOT)
    from erfnet_pytorch_amd.ddp import GradientReducer
    x, y = _batch(0)
    results = []
    for flat in (False, True):
        model = _model()
        red = GradientReducer(model, bucket_bytes=128 << 10)
        # set-up pass without an update: discovers the live parameters
        red.zero_grad()
        torch.nn.functional.l1_loss(model(x), y).backward()
        red.finish()
        params = red.flatten_parameters() if flat else [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, 1e-3, eps=1e-4, weight_decay=1e-6)
        for _ in range(3):
            red.zero_grad()
            torch.nn.functional.l1_loss(model(x), y).backward()
            red.finish()
            opt.step()
        results.append({k: p.detach().clone() for k, p in model.named_parameters()})
        if flat:
            assert len(params) == len(red.bucket_sizes) < len(list(model.parameters())) // 4
    a, b = results
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_direct_written_gradients_still_launch_buckets_from_inside_backward():
    """With direct_write the kernels write a weight gradient into its bucket slice and hand autograd
    None; the reducer counts that parameter in the post-accumulate hook autograd still runs for it.
    If a torch version stopped running the hook for an undefined gradient, every bucket would leave
    from finish() only — correct, but without any overlap with backward, and unnoticed.  Here: a
    layer that delivers its gradients exactly the way the HIP blocks do (functional._grad_targets /
    _deliver), a one-rank gloo group with reduce_single_rank, and the assertion that the buckets'
    all-reduces were launched before finish()."""
    sys.path.insert(0, ROOT)
    from erfnet_pytorch_amd import functional as hipf
    from erfnet_pytorch_amd.ddp import GradientReducer

    class DirectLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            ctx.params = (w, b)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, gy):
            x, w = ctx.saved_tensors
            tg = hipf._grad_targets(ctx.params, (w.shape, (w.shape[0],)), x.device)
            tg[0][0].copy_(gy.t() @ x)              # what a kernel does: write into the buffer it was given
            tg[1][0].copy_(gy.sum(0))
            dw, db = hipf._deliver(tg)
            return gy @ w, dw, db

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.Linear(32, 32) for _ in range(6)])

        def forward(self, x):
            for l in self.layers:
                x = torch.relu(DirectLinear.apply(x, l.weight, l.bias))
            return x

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        model = Net()
        ref = Net()
        ref.load_state_dict(model.state_dict())
        red = GradientReducer(model, bucket_bytes=2 * (32 * 32 + 32) * 4, direct_write=True, reduce_single_rank=True)
        x = torch.rand(8, 32)
        for step in range(3):
            red.zero_grad()
            model(x).sum().backward()
            if step > 0:
                # every bucket complete -> launched from the hook, i.e. before finish()
                assert all(v == -1 for v in red._pending.values()), (step, red._pending)
                assert len(red._written) == 12              # all weights and biases took the direct path
            red.finish()
        for p in ref.parameters():
            p.grad = None
        y = x
        for l in ref.layers:
            y = torch.relu(l(y))
        y.sum().backward()
        for (k, p), q in zip(model.named_parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-5), k
    finally:
        dist.destroy_process_group()
