"""Regression / consistency losses (train/evaluation_functions.py:32-131): the numpy oracle and the
CPU branch of the modules against the goldens (values and gradients produced by the reference's own
classes, tests/golden/make_golden_extra.py), and the HIP kernels against both."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import oracle as orc
from erfnet_pytorch_amd import evaluation_functions as ef

HERE = os.path.dirname(os.path.abspath(__file__))
EXTRA = np.load(os.path.join(HERE, "golden", "extra.npz"))
with open(os.path.join(HERE, "golden", "extra.json")) as f:
    META = json.load(f)["losses"]
N, K, H, W = META["shape"]


def _softmax(x):
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def golden_inputs():
    """The seeded inputs of make_golden_extra.gen_losses."""
    d = dict(targets=EXTRA["losses/targets"],
             l1_out=gu.det_uniform("loss.l1.out", (N, 1, H, W), 0.0, 3.0),
             ll_cost=gu.det_uniform("loss.ll.cost", (N, H, W), 0.0, 3.0),
             ll_var=gu.det_uniform("loss.ll.var", (N, H, W), 0.05, 2.0),
             cp_cost=gu.det_uniform("loss.cp.cost", (1, K, 1, 1), 0.0, 3.0),
             cp_var=gu.det_uniform("loss.cp.var", (1, K, 1, 1), 0.05, 2.0),
             mse_a=gu.det_uniform("loss.mse.a", (N, K, H, W), 0.0, 1.0),
             mse_b=gu.det_uniform("loss.mse.b", (N, K, H, W), 0.0, 1.0))
    d["prob"] = torch.softmax(torch.from_numpy(gu.det_normal("loss.cp.logits", (N, K, H, W))), dim=1).numpy()
    return d


# name -> (module factory, input keys (last = targets), golden gradient keys per differentiable input, oracle call)
CASES = {
    "l1_masked": (lambda: ef.L1LossMasked(), ("l1_out",), ("d_out",),
                  lambda d: orc.masked_loss("l1", d["l1_out"], None, None, d["targets"])),
    "ll_masked": (lambda: ef.LogLikelihoodLossMasked(opt_eps=META["opt_eps"]), ("ll_cost", "ll_var"),
                  ("d_cost", "d_var"),
                  lambda d: orc.masked_loss("ll", d["ll_cost"][:, None], d["ll_var"], None, d["targets"], META["opt_eps"])),
    "l1_classprob": (lambda: ef.L1LossClassProbMasked(), ("prob", "cp_cost"), ("d_prob", "d_cost"),
                     lambda d: orc.masked_loss("l1cp", d["prob"], d["cp_cost"], None, d["targets"])),
    "ll_classprob": (lambda: ef.LogLikelihoodLossClassProbMasked(), ("prob", "cp_cost", "cp_var"),
                     ("d_prob", "d_cost", "d_var"),
                     lambda d: orc.masked_loss("llcp", d["prob"], d["cp_cost"], d["cp_var"], d["targets"])),
}


def _close(got, want, tol=2e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape or got.size == want.size, (got.shape, want.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got.reshape(-1) - want.reshape(-1)).max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_goldens(name):
    _, _, gkeys, call = CASES[name]
    d = golden_inputs()
    loss, da, db, dc = call(d)
    assert abs(loss - META["values"][name]) <= 2e-6 * abs(META["values"][name])
    grads = [g for g in (da, db, dc) if g is not None]
    for g, key in zip(grads, gkeys):
        _close(g, EXTRA[f"losses/{name}/{key}"])


def test_oracle_mse_weighted_matches_reference_golden():
    d = golden_inputs()
    v, g = orc.mse_weighted(d["mse_a"], d["mse_b"], META["scale_factor"], META["weight"])
    assert abs(v - META["values"]["mse_weighted"]) <= 2e-6 * v
    _close(g, EXTRA["losses/mse_weighted/d_out"])


def _run_module(name, device):
    make, ikeys, gkeys, _ = CASES[name]
    d = golden_inputs()
    ins = [torch.from_numpy(d[k]).to(device).requires_grad_(True) for k in ikeys]
    loss = make()(*ins, torch.from_numpy(d["targets"]).to(device))
    loss.backward()
    return float(loss.detach()), [i.grad.cpu().numpy() for i in ins], gkeys


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpu_modules_match_reference_goldens(name):
    loss, grads, gkeys = _run_module(name, "cpu")
    assert abs(loss - META["values"][name]) <= 2e-6 * abs(META["values"][name])
    for g, key in zip(grads, gkeys):
        _close(g, EXTRA[f"losses/{name}/{key}"])


def test_cpu_mse_weighted_and_l1():
    d = golden_inputs()
    a = torch.from_numpy(d["mse_a"]).requires_grad_(True)
    loss = ef.MSELossWeighted(META["scale_factor"])(a, torch.from_numpy(d["mse_b"]), META["weight"])
    loss.backward()
    assert abs(float(loss.detach()) - META["values"]["mse_weighted"]) <= 2e-6 * META["values"]["mse_weighted"]
    _close(a.grad.numpy(), EXTRA["losses/mse_weighted/d_out"])
    x, y = torch.from_numpy(d["mse_a"]), torch.from_numpy(d["mse_b"])
    assert float(ef.L1Loss()(x, y)) == float((x - y).abs().mean())


def test_cpu_mean_accuracy_histograms():
    """MeanAccuracy / ClassificationAccuracy on CPU tensors against a per-class count in numpy
    (train/evaluation_functions.py:133-163): absent classes give nan and leave the mean alone."""
    rng = np.random.default_rng(5)
    prob = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
    tgt = rng.integers(-1, 5, (2, 9, 11))           # class 5 never labelled, -1 = unlabelled
    tgt[tgt == 3] = 2                               # class 3 absent as well
    pred = prob.argmax(1)
    acc = [((pred == c) & (tgt == c)).sum() / (tgt == c).sum() if (tgt == c).any() else np.nan for c in range(6)]
    mean, per_class = ef.MeanAccuracy(6)(torch.from_numpy(prob), torch.from_numpy(tgt))
    np.testing.assert_allclose(per_class, np.array(acc, np.float32), rtol=1e-6, equal_nan=True)
    assert abs(float(mean) - np.nanmean(acc)) < 1e-6
    total = float(ef.ClassificationAccuracy()(torch.from_numpy(prob), torch.from_numpy(tgt)))
    assert abs(total - (pred == tgt).sum() / (tgt >= 0).sum()) < 1e-6


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_losses_match_reference_goldens(name):
    loss, grads, gkeys = _run_module(name, "cuda")
    assert abs(loss - META["values"][name]) <= 1e-5 * abs(META["values"][name])
    for g, key in zip(grads, gkeys):
        _close(g, EXTRA[f"losses/{name}/{key}"], tol=1e-4)


@pytest.mark.gpu
def test_hip_mse_weighted_matches_reference_golden():
    d = golden_inputs()
    a = torch.from_numpy(d["mse_a"]).cuda().requires_grad_(True)
    loss = ef.MSELossWeighted(META["scale_factor"])(a, torch.from_numpy(d["mse_b"]).cuda(), META["weight"])
    loss.backward()
    assert abs(float(loss.detach()) - META["values"]["mse_weighted"]) <= 1e-5 * META["values"]["mse_weighted"]
    _close(a.grad.cpu().numpy(), EXTRA["losses/mse_weighted/d_out"], tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 1, 1), (3, 20, 37, 53), (2, 7, 128, 257), (1, 256, 5, 9)])
@pytest.mark.parametrize("kind", ["l1", "ll", "l1cp", "llcp"])
def test_hip_losses_against_oracle_ragged(kind, shape):
    """Sizes that do not fill the last block, one pixel, the largest class count."""
    from erfnet_pytorch_amd import functional as F
    n, k, h, w = shape
    rng = np.random.default_rng(hash((kind, shape)) % (1 << 32))
    tgt = rng.uniform(0, 3, (n, h, w)).astype(np.float32)
    tgt[rng.uniform(size=tgt.shape) < 0.4] = -1.0
    if n * h * w == 1:
        tgt[...] = 1.25
    eps = 1e-4
    if kind in ("l1", "ll"):
        a = rng.uniform(0, 3, (n, 1, h, w)).astype(np.float32)
        b = rng.uniform(0.05, 2, (n, h, w)).astype(np.float32) if kind == "ll" else None
        c = None
    else:
        a = _softmax(rng.standard_normal((n, k, h, w))).astype(np.float32)
        b = rng.uniform(0, 3, k).astype(np.float32)
        c = rng.uniform(0.05, 2, k).astype(np.float32) if kind == "llcp" else None
    want = orc.masked_loss(kind, a, b, c, tgt, eps)
    code = dict(l1=F.ML_L1, ll=F.ML_LL, l1cp=F.ML_L1CP, llcp=F.ML_LLCP)[kind]
    ts = [None if v is None else torch.from_numpy(v).cuda().requires_grad_(True) for v in (a, b, c)]
    loss = F.masked_loss(code, ts[0], ts[1], ts[2], torch.from_numpy(tgt).cuda(), eps)
    loss.backward()
    assert abs(float(loss) - want[0]) <= 2e-5 * abs(want[0])
    for tns, g in zip(ts, want[1:]):
        if tns is not None:
            _close(tns.grad.cpu().numpy(), g, tol=2e-4)


@pytest.mark.gpu
def test_hip_masked_loss_without_labels_is_nan_like_the_reference():
    x = torch.rand(2, 1, 8, 8, device="cuda", requires_grad=True)
    tgt = torch.full((2, 8, 8), -1.0, device="cuda")
    loss = ef.L1LossMasked()(x, tgt)
    assert torch.isnan(loss)
    assert torch.isnan(ef.L1LossMasked()(x.detach().cpu(), tgt.cpu()))
