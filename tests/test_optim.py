"""Parameter updates: torch.optim.Adam (train/main.py:213,381), torch.optim.SGD (imagenet/main.py:83-85,187)
and the mean-teacher update (train/main.py:140-142).

CPU: the numpy oracle against fixtures written by stock torch (tests/golden/make_golden_optim.py).
GPU: erf_adam_step / erf_sgd_step / erf_ema_update through the C ABI and through the optimiser classes
against the oracle and the same fixtures.  Tolerance: the update is a handful of fp32 operations per
element whose contraction into fused multiply-adds differs between compilers, so parameters agree to
2 ulp-sized steps of the update (1e-6 of the largest update) and the moments to 1e-6 relative; the
mean-teacher update has one rounding order only and must be bit-identical."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden_optim as mk  # noqa: E402  (sizes, cases and seeded inputs only; writes nothing)
from oracle import optim_oracle as oo  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim.npz"))
DEV = "cuda:0"


def close(got, ref, rel, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    tol = rel * max(1e-30, np.abs(ref).max())
    err = np.abs(got - ref).max()
    assert err <= tol, (what, err, tol)


# ------------------------------------------------------------------ oracle against stock torch
@pytest.mark.parametrize("name", list(mk.ADAM_CASES))
def test_oracle_adam_matches_torch(name):
    hp = mk.ADAM_CASES[name]
    ps = mk.params(f"adam.{name}")
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    for step in range(mk.STEPS):
        for i, p in enumerate(ps):
            oo.adam_step(p, mk.grad(f"adam.{name}", step, i, p.size), ms[i], vs[i], step + 1, hp["lr"],
                         hp["betas"][0], hp["betas"][1], hp["eps"], hp["weight_decay"])
            if step in mk.KEPT:
                close(p, GOLD[f"adam/{name}/s{step}/p{i}"], 3e-7, ("p", step, i))
                close(ms[i], GOLD[f"adam/{name}/s{step}/m{i}"], 1e-6, ("m", step, i))
                close(vs[i], GOLD[f"adam/{name}/s{step}/v{i}"], 1e-6, ("v", step, i))


@pytest.mark.parametrize("name", list(mk.SGD_CASES))
def test_oracle_sgd_matches_torch(name):
    hp = mk.SGD_CASES[name]
    ps = mk.params(f"sgd.{name}")
    bufs = [np.zeros_like(p) for p in ps]
    for step in range(mk.STEPS):
        for i, p in enumerate(ps):
            oo.sgd_step(p, mk.grad(f"sgd.{name}", step, i, p.size), bufs[i], step == 0, hp["lr"], hp["momentum"],
                        hp["weight_decay"])
            if step in mk.KEPT:
                close(p, GOLD[f"sgd/{name}/s{step}/p{i}"], 3e-7, ("p", step, i))
                close(bufs[i], GOLD[f"sgd/{name}/s{step}/m{i}"], 1e-6, ("buf", step, i))


@pytest.mark.parametrize("d", mk.EMA_DISCOUNTS)
def test_oracle_ema_matches_reference_formula(d):
    tgt, src = mk.params(f"ema.{d}.t"), mk.params(f"ema.{d}.s")
    for step in range(2):
        for i in range(len(tgt)):
            oo.ema_update(tgt[i], src[i], d)
            assert np.array_equal(tgt[i], GOLD[f"ema/{d}/s{step}/t{i}"]), (step, i)


def test_optimisers_refuse_host_tensors():
    """No host path: a parameter that is not on a GPU is an error, not a silent torch fallback."""
    from erfnet_pytorch_amd import optim
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    for opt in (optim.Adam([p], 1e-3), optim.SGD([p], 0.1, momentum=0.9)):
        with pytest.raises(RuntimeError):
            opt.step()
    with pytest.raises(ValueError):
        optim.Adam([p], 1e-3, betas=(1.0, 0.9))


# ------------------------------------------------------------------ HIP kernels
def _tables(lists):
    n = len(lists[0])
    return [(C.c_void_p * n)(*[t.data_ptr() for t in lst]) for lst in lists], \
        (C.c_longlong * n)(*[t.numel() for t in lists[0]])


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.ADAM_CASES))
def test_adam_c_abi_against_golden(name):
    from erfnet_pytorch_amd import _lib
    hp = mk.ADAM_CASES[name]
    ps = [torch.from_numpy(a).to(DEV) for a in mk.params(f"adam.{name}")]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    n0 = _lib.launch_count()
    for step in range(mk.STEPS):
        gs = [torch.from_numpy(mk.grad(f"adam.{name}", step, i, p.numel())).to(DEV) for i, p in enumerate(ps)]
        ptrs, numel = _tables([ps, gs, ms, vs])
        _lib.check(_lib.lib().erf_adam_step(len(ps), *ptrs, numel, hp["lr"], hp["betas"][0], hp["betas"][1],
                                            hp["eps"], hp["weight_decay"], step + 1, _stream()))
        if step in mk.KEPT:
            for i in range(len(ps)):
                close(ps[i].cpu().numpy(), GOLD[f"adam/{name}/s{step}/p{i}"], 3e-7, ("p", step, i))
                close(ms[i].cpu().numpy(), GOLD[f"adam/{name}/s{step}/m{i}"], 1e-6, ("m", step, i))
                close(vs[i].cpu().numpy(), GOLD[f"adam/{name}/s{step}/v{i}"], 1e-6, ("v", step, i))
    assert _lib.launch_count() == n0 + mk.STEPS          # one launch per step for the whole list


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.SGD_CASES))
def test_sgd_c_abi_against_golden(name):
    from erfnet_pytorch_amd import _lib
    hp = mk.SGD_CASES[name]
    ps = [torch.from_numpy(a).to(DEV) for a in mk.params(f"sgd.{name}")]
    bufs = [torch.zeros_like(p) for p in ps]
    for step in range(mk.STEPS):
        gs = [torch.from_numpy(mk.grad(f"sgd.{name}", step, i, p.numel())).to(DEV) for i, p in enumerate(ps)]
        ptrs, numel = _tables([ps, gs, bufs])
        _lib.check(_lib.lib().erf_sgd_step(len(ps), *ptrs, numel, hp["lr"], hp["momentum"], hp["weight_decay"],
                                           _stream()))
        if step in mk.KEPT:
            for i in range(len(ps)):
                close(ps[i].cpu().numpy(), GOLD[f"sgd/{name}/s{step}/p{i}"], 3e-7, ("p", step, i))
                close(bufs[i].cpu().numpy(), GOLD[f"sgd/{name}/s{step}/m{i}"], 1e-6, ("buf", step, i))


@pytest.mark.gpu
def test_updates_ragged_lists_against_oracle():
    """More tensors than one launch holds (70), empty tensors, sizes around the block's chunk (4096),
    tensors that start 4 bytes off a 16-byte boundary, and a large one; Adam, SGD and the mean-teacher
    update against the oracle on the same seeded values."""
    from erfnet_pytorch_amd import _lib
    sizes = [4095, 4096, 4097, 0, 8192, 1, 300001] + [((7 * i) % 61) + 1 for i in range(63)]
    assert len(sizes) == 70
    host = {k: [gu.det_normal(f"optim.rag.{k}.{i}", (n,)).astype(np.float32) for i, n in enumerate(sizes)]
            for k in "pgsq"}
    for a in host["g"]:
        a *= 0.1

    def dev(arrs):
        # every other tensor lives one element into its allocation: not 16-byte aligned
        out = []
        for i, a in enumerate(arrs):
            off = i % 2
            buf = torch.zeros(a.size + off, device=DEV)
            buf[off:] = torch.from_numpy(a)
            out.append(buf[off:])
        return out

    lib = _lib.lib()
    # Adam, three steps with a fixed gradient
    p, g = dev(host["p"]), dev(host["g"])
    m, v = dev([np.zeros_like(a) for a in host["p"]]), dev([np.zeros_like(a) for a in host["p"]])
    ptrs, numel = _tables([p, g, m, v])
    n0 = _lib.launch_count()
    for step in range(3):
        _lib.check(lib.erf_adam_step(len(p), *ptrs, numel, 5e-4, 0.9, 0.999, 1e-8, 1e-4, step + 1, _stream()))
    assert _lib.launch_count() == n0 + 3 * 3            # 69 non-empty tensors, 32 per launch
    rp = [a.copy() for a in host["p"]]
    rm = [np.zeros_like(a) for a in rp]
    rv = [np.zeros_like(a) for a in rp]
    for step in range(3):
        for i in range(len(rp)):
            oo.adam_step(rp[i], host["g"][i], rm[i], rv[i], step + 1, 5e-4, 0.9, 0.999, 1e-8, 1e-4)
    for i, n in enumerate(sizes):
        if n:
            close(p[i].cpu().numpy(), rp[i], 3e-7, ("adam p", i))
            close(m[i].cpu().numpy(), rm[i], 1e-6, ("adam m", i))
            close(v[i].cpu().numpy(), rv[i], 1e-6, ("adam v", i))
    # SGD, three steps
    p, buf = dev(host["p"]), dev([np.zeros_like(a) for a in host["p"]])
    ptrs, numel = _tables([p, g, buf])
    for step in range(3):
        _lib.check(lib.erf_sgd_step(len(p), *ptrs, numel, 0.1, 0.9, 1e-4, _stream()))
    rp = [a.copy() for a in host["p"]]
    rb = [np.zeros_like(a) for a in rp]
    for step in range(3):
        for i in range(len(rp)):
            oo.sgd_step(rp[i], host["g"][i], rb[i], step == 0, 0.1, 0.9, 1e-4)
    for i, n in enumerate(sizes):
        if n:
            close(p[i].cpu().numpy(), rp[i], 3e-7, ("sgd p", i))
            close(buf[i].cpu().numpy(), rb[i], 1e-6, ("sgd buf", i))
    # mean-teacher update: bit-identical
    tgt, src = dev(host["s"]), dev(host["q"])
    ptrs, numel = _tables([tgt, src])
    _lib.check(lib.erf_ema_update(len(tgt), *ptrs, numel, 0.01, _stream()))
    for i, n in enumerate(sizes):
        ref = host["s"][i].copy()
        oo.ema_update(ref, host["q"][i], 0.01)
        assert np.array_equal(tgt[i].cpu().numpy(), ref), ("ema", i)
    # nothing beside the tensors was touched: the element in front of an offset tensor is still zero
    torch.cuda.synchronize()
    for lst in (p, buf, tgt):
        for i, x in enumerate(lst):
            if i % 2:
                assert x.storage_offset() == 1 and float(x._base[0] if x._base is not None else 0.0) == 0.0


@pytest.mark.gpu
def test_update_entry_points_reject_bad_arguments():
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    p = torch.zeros(8, device=DEV)
    ptrs, numel = _tables([[p], [p], [p], [p]])
    assert lib.erf_adam_step(1, *ptrs, numel, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, _stream()) != 0      # step 0
    assert b"step" in lib.erf_last_error()
    assert lib.erf_adam_step(1, *ptrs, numel, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, _stream()) != 0      # beta1 = 1
    assert lib.erf_adam_step(1, ptrs[0], None, ptrs[2], ptrs[3], numel, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1,
                             _stream()) != 0
    null = (C.c_void_p * 1)(None)
    assert lib.erf_sgd_step(1, ptrs[0], null, ptrs[2], numel, 0.1, 0.9, 0.0, _stream()) != 0
    assert lib.erf_sgd_step(1, ptrs[0], ptrs[1], ptrs[2], numel, -0.1, 0.9, 0.0, _stream()) != 0
    assert lib.erf_ema_update(0, None, None, None, 0.5, _stream()) == 0                            # empty list
    torch.cuda.synchronize()
    assert float(p.abs().sum()) == 0.0


@pytest.mark.gpu
def test_optimiser_classes_follow_torch():
    """optim.Adam / optim.SGD against torch.optim on the GPU over five steps of a small model with a
    learning-rate schedule, a parameter that gets its first gradient late, and a state_dict round trip
    in both directions (checkpoints are interchangeable with torch's)."""
    from erfnet_pytorch_amd import optim
    for kind in ("adam", "sgd"):
        torch.manual_seed(3)
        base = [torch.randn(n, device=DEV) for n in (5000, 17, 4096, 1)]

        def make(cls, **hp):
            ps = [torch.nn.Parameter(b.clone()) for b in base]
            return ps, cls(ps, **hp)

        if kind == "adam":
            hp = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
            pa, oa = make(optim.Adam, **hp)
            pb, ob = make(torch.optim.Adam, **hp)
        else:
            hp = dict(lr=0.1, momentum=0.9, weight_decay=1e-4)
            pa, oa = make(optim.SGD, **hp)
            pb, ob = make(torch.optim.SGD, **hp)
        sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda e: 0.9 ** e)      # train/main.py:250
        sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda e: 0.9 ** e)
        for step in range(5):
            if step == 3:       # checkpoint hand-over: each optimiser resumes from the other's state
                da, db = oa.state_dict(), ob.state_dict()
                oa.load_state_dict(db)
                ob.load_state_dict(da)
            for i in range(len(base)):
                g = torch.randn_like(base[i]) * 0.1
                late = i == 1 and step < 2          # this parameter has no gradient in the first two steps
                pa[i].grad = None if late else g.clone()
                pb[i].grad = None if late else g.clone()
            oa.step()
            ob.step()
            sa.step()
            sb.step()
            for i in range(len(base)):
                close(pa[i].detach().cpu().numpy(), pb[i].detach().cpu().numpy(), 3e-7, (kind, step, i))
        for i in range(len(base)):
            if kind == "adam":
                assert int(oa.state[pa[i]]["step"]) == int(ob.state[pb[i]]["step"]) == (3 if i == 1 else 5)
                close(oa.state[pa[i]]["exp_avg_sq"].cpu().numpy(), ob.state[pb[i]]["exp_avg_sq"].cpu().numpy(), 1e-6,
                      ("v", i))
            else:
                close(oa.state[pa[i]]["momentum_buffer"].cpu().numpy(),
                      ob.state[pb[i]]["momentum_buffer"].cpu().numpy(), 1e-6, ("buf", i))


@pytest.mark.gpu
def test_training_step_with_library_adam_matches_reference():
    """The reference's two training steps (tests/golden/trainstep.json: CrossEntropy(ignore -1) +
    Adam(lr 1e-4, eps 1e-4, wd 1e-6), train/main.py:43-51,213,297-392) with erfnet_pytorch_amd.optim.Adam
    as the optimiser, and the mean-teacher update of a GPU model pair against the formula."""
    from erfnet_pytorch_amd import mean_teacher, optim
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    from test_hip_parity import make_model
    ref = gu.load_json("trainstep.json")
    net = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV).train()
    teacher = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV)
    before = [p.detach().clone() for p in teacher.parameters()]
    opt = optim.Adam(net.parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)
    crit = CrossEntropyLoss2d()
    x = torch.from_numpy(gu.det_uniform("trainstep.x", (2, 3, 64, 128), 0.0, 1.0)).to(DEV)
    lab = torch.from_numpy(gu.det_labels("trainstep.lab", (2, 64, 128), -1, 20)).to(DEV)
    for step in range(2):
        opt.zero_grad()
        out, _ = net(x)
        loss = crit(out, lab)
        loss.backward()
        opt.step()
        assert abs(loss.item() - ref["losses"][step]) < 1e-3, (step, loss.item(), ref["losses"][step])
    for k, p in net.named_parameters():
        s, a = ref["params"][k]
        pd = p.detach().double()
        assert abs(float(pd.sum()) - s) <= 2e-4 * max(1.0, a) + 1e-5, k
        assert abs(float(pd.abs().sum()) - a) <= 2e-4 * max(1.0, a) + 1e-5, k
    mean_teacher.copyWeightsToModelWithDiscount(net, teacher, 0.01)
    for b, s, tp in zip(before, net.parameters(), teacher.parameters()):
        assert torch.equal(tp.detach(), b * (1 - 0.01) + 0.01 * s.detach())       # train/main.py:142


@pytest.mark.gpu
def test_mean_teacher_update_reaches_a_bf16_teacher():
    """A teacher kept in bf16 (model.bfloat16()) is served through fp32 copies of its parameters; the
    mean-teacher update (train/main.py:140-142,390) must be visible to the very next forward — the
    copies are refreshed, not reused."""
    from erfnet_pytorch_amd import mean_teacher
    from test_hip_parity import make_model
    student = gu.fill_state_dict(make_model("net_classification_20"), tag="mt.student").to(DEV).eval()
    teacher = gu.fill_state_dict(make_model("net_classification_20"), tag="mt.teacher").to(DEV).eval().bfloat16()
    x = torch.from_numpy(gu.det_uniform("mt.x", (1, 3, 64, 128), 0.0, 1.0)).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        y0 = teacher(x)[0].float().clone()
        mean_teacher.copyWeightsToModelWithDiscount(student, teacher, 0.5)
        y1 = teacher(x)[0].float().clone()
        # the same weights in a fresh model: nothing cached can be involved
        fresh = make_model("net_classification_20").to(DEV).eval().bfloat16()
        fresh.load_state_dict(teacher.state_dict())
        y2 = fresh(x)[0].float()
    assert float((y1 - y0).abs().max()) > 1e-3 * float(y0.abs().max()), "the update did not reach the forward pass"
    assert torch.equal(y1, y2)
