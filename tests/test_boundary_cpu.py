"""CPU-side checks of the drop-in boundary: module tree / state_dict surface, the host plumbing path
on CPU tensors against the golden vectors, and the C-ABI library's exported symbols."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

import golden_util as gu

MAN = gu.load_json("manifest.json")
NETS = gu.load_npz("nets.npz")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODELS = {
    "eval_erfnet_20": ("erfnet", "ERFNet", (20,), {}),
    "eval_erfnet_nobn_19": ("erfnet_nobn", "ERFNet", (19,), {}),
    "net_classification_20": ("net_classification", "Net", (), dict(softmax_classes=20)),
    "net_regression_0": ("net_regression", "Net", (), {}),
    "net_regression_0_lik": ("net_regression", "Net", (), dict(likelihood_loss=True)),
    "net_regression_20": ("net_regression", "Net", (), dict(softmax_classes=20)),
    "net_regression_small_0": ("net_regression_small", "Net", (), {}),
    "net_regression_ladder_0": ("net_regression_ladder", "Net", (), {}),
    "net_regression_ladder_small_0": ("net_regression_ladder_small", "Net", (), {}),
    "net_classification_ladder_20": ("net_classification_ladder", "Net", (), dict(softmax_classes=20)),
    "imagenet_1000": ("erfnet_imagenet", "ERFNet", (1000,), {}),
}


def make_model(key):
    mod, cls, args, kw = MODELS[key]
    m = importlib.import_module("erfnet_pytorch_amd." + mod)
    return getattr(m, cls)(*args, **kw)


def first(o):
    return o[0] if isinstance(o, (tuple, list)) else o


@pytest.mark.parametrize("key", list(MODELS))
def test_state_dict_surface(key):
    """Same keys, same order, same shapes and parameter count as the reference's model."""
    net = make_model(key)
    got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    assert got == MAN["nets"][key]["state_dict"]
    assert sum(p.numel() for p in net.parameters()) == MAN["nets"][key]["n_params"]


def test_upstream_alias_and_attributes():
    from erfnet_pytorch_amd import erfnet, erfnet_imagenet, net_classification
    assert erfnet.Net is erfnet.ERFNet
    n = net_classification.Net(softmax_classes=20)
    assert n.class_power.shape == (1, 20, 1, 1)
    assert hasattr(n.encoder, "initial_block") and hasattr(n.decoder, "scalar_output_conv")
    # hand-off used by the reference's train/main.py:857-860
    assert hasattr(erfnet_imagenet.ERFNet(10).features, "encoder")


@pytest.mark.parametrize("key", [k for k in MODELS if k != "imagenet_1000"])
def test_cpu_plumbing_small(key):
    """CPU tensors run through stock torch ops on the same parameters: logits equal the reference's."""
    net = gu.fill_state_dict(make_model(key), tag=key).eval()
    x = torch.from_numpy(gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0))
    with torch.no_grad():
        y = first(net(x)).numpy()
    gu.check(NETS, f"{key}/small/y", y, 1e-5, 1e-5)
    assert np.array_equal(y.argmax(1).astype(np.uint8), NETS[f"{key}/small/argmax"])


def test_cpu_forward_512x1024_matches_reference():
    """BASELINE configs[0]: Net(20 classes) forward on CPU, 1x3x512x1024, vs the reference."""
    key = "eval_erfnet_20"
    net = gu.fill_state_dict(make_model(key), tag=key).eval()
    x = torch.from_numpy(gu.det_uniform("net.x_big", (1, 3, 512, 1024), 0.0, 1.0))
    with torch.no_grad():
        y = net(x).numpy()
    gu.check(NETS, f"{key}/big/y", y, 1e-3, 1e-3)
    np.testing.assert_allclose(y[:, :, 224:288, 480:544], NETS[f"{key}/big/crop"], atol=1e-3)
    assert np.array_equal(y.argmax(1).astype(np.uint8), NETS[f"{key}/big/argmax"])


def test_only_encode_cpu():
    key = "eval_erfnet_20"
    net = gu.fill_state_dict(make_model(key), tag=key).eval()
    x = torch.from_numpy(gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0))
    with torch.no_grad():
        y = net(x, only_encode=True).numpy()
    assert y.shape == (1, 20, 8, 16)
    gu.check(NETS, "eval_erfnet_20/small/y_encode", y, 1e-5, 1e-5)


def test_odd_size_raises_on_cpu():
    from erfnet_pytorch_amd.erfnet_blocks import DownsamplerBlock
    with pytest.raises(RuntimeError):
        DownsamplerBlock(3, 16)(torch.zeros(1, 3, 15, 24))


def test_iou_eval_golden():
    """The reference's iouEval algorithm (eval/iouEval.py:21-75) restated with a confusion count:
    per-class IoU = tp / (tp + fp + fn), class 19 ignored; pinned by the reference's own output."""
    pred = gu.det_labels("iou.pred", (2, 1, 32, 48), 0, 20).reshape(-1)
    lab = gu.det_labels("iou.lab", (2, 1, 32, 48), 0, 20).reshape(-1)
    keep = lab != 19
    ious = []
    for c in range(19):
        tp = np.sum((pred == c) & (lab == c))
        fp = np.sum((pred == c) & (lab != c) & keep)
        fn = np.sum((pred != c) & (lab == c))
        ious.append(tp / (tp + fp + fn + 1e-15))
    ref = gu.load_json("iou.json")
    np.testing.assert_allclose(ious, ref["per_class"], rtol=1e-12)
    assert abs(np.mean(ious) - ref["mean"]) < 1e-12


# ------------------------------------------------------------------------------- C-ABI library
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "erfnet_hip.h")).read()
    return sorted(set(re.findall(r"\b(erf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """liberfnet_hip.so loads without a GPU and exports what include/erfnet_hip.h declares."""
    from erfnet_pytorch_amd import _lib
    so = _lib.build()
    handle = ctypes.CDLL(so)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(handle, name), f"{name} is declared in the header but not exported"
    assert sorted(_lib.EXPORTS) == declared
    handle.erf_version.restype = ctypes.c_int
    assert handle.erf_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    from erfnet_pytorch_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_SO", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="missing"):
        _lib.lib()


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under erfnet_pytorch_amd/ may reference it."""
    pkg = os.path.join(ROOT, "erfnet_pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("test infrastructure", ""), f


def test_cross_entropy_loss2d_cpu_matches_torch():
    """CrossEntropyLoss2d keeps the reference's signature (weight accepted and ignored,
    train/evaluation_functions.py:8-11) and its CPU path is torch's own op."""
    import torch
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 6, 7, generator=g)
    lab = torch.randint(-1, 5, (2, 6, 7), generator=g)
    a = CrossEntropyLoss2d(weight=torch.ones(5))(x, lab)
    b = torch.nn.functional.cross_entropy(x, lab, ignore_index=-1)
    assert torch.equal(a, b)


def test_iou_eval_class_cpu_matches_golden():
    """The drop-in iouEval class on CPU label maps (the reference's one-hot formulation path)
    reproduces the reference's own result; counters are CPU double tensors as in the reference."""
    import torch
    from erfnet_pytorch_amd.iouEval import iouEval, getColorEntry, colors
    pred = torch.from_numpy(gu.det_labels("iou.pred", (2, 1, 32, 48), 0, 20).astype(np.int64))
    lab = torch.from_numpy(gu.det_labels("iou.lab", (2, 1, 32, 48), 0, 20).astype(np.int64))
    ev = iouEval(20)
    ev.addBatch(pred[:1], lab[:1])
    ev.addBatch(pred[1:], lab[1:])
    mean, per_class = ev.getIoU()
    ref = gu.load_json("iou.json")
    np.testing.assert_allclose(per_class.numpy(), ref["per_class"], rtol=1e-12)
    assert abs(float(mean) - ref["mean"]) < 1e-12
    assert ev.tp.dtype == torch.float64 and ev.tp.shape == (19,)
    assert iouEval(5, ignoreIndex=19).ignoreIndex == -1
    assert getColorEntry(0.1) == colors.RED and getColorEntry(0.95) == colors.GREEN and getColorEntry("x") == colors.ENDC


def test_accuracy_metrics_cpu():
    """ClassificationAccuracy / MeanAccuracy on CPU tensors follow the reference's torch formulation."""
    import torch
    from erfnet_pytorch_amd.evaluation_functions import ClassificationAccuracy, MeanAccuracy
    prob = torch.tensor([[[[0.9, 0.2]], [[0.1, 0.8]], [[0.0, 0.0]]]])          # [1,3,1,2]: argmax = 0, 1
    tgt = torch.tensor([[[0, 2]]])
    assert float(ClassificationAccuracy()(prob, tgt)) == 0.5
    mean_acc, class_acc = MeanAccuracy(3)(prob, tgt)
    assert float(mean_acc) == 0.5 and class_acc[0] == 1.0 and np.isnan(class_acc[1]) and class_acc[2] == 0.0


def test_mean_teacher_update_matches_reference_formula():
    """copyWeightsToModelWithDiscount (multi-tensor form) against the per-parameter formula of
    train/main.py:140-142; copyWeightsToModelNoGrad only freezes the teacher."""
    import torch
    from erfnet_pytorch_amd import mean_teacher, net_regression_small
    torch.manual_seed(1)
    student, teacher = net_regression_small.Net(), net_regression_small.Net()
    with torch.no_grad():
        for p in teacher.parameters():
            p.add_(0.25)
    want = [t.data * (1 - 0.01) + 0.01 * s.data for s, t in zip(student.parameters(), teacher.parameters())]
    mean_teacher.copyWeightsToModelNoGrad(student, teacher)
    assert not any(p.requires_grad for p in teacher.parameters())
    mean_teacher.copyWeightsToModelWithDiscount(student, teacher, 0.01)
    for w, t in zip(want, teacher.parameters()):
        assert torch.allclose(w, t.data, rtol=1e-6, atol=1e-8)
