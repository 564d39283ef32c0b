"""Pins the plain-C CPU oracle (oracle/erfnet_oracle.c) against the golden vectors that were
produced by importing the reference's own modules (tests/golden/make_golden.py)."""
import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as orc

MAN = gu.load_json("manifest.json")
BLK = gu.load_npz("blocks.npz")
CASES = {c["name"]: c for c in MAN["block_cases"]}

# fp32 accumulation-order differences between the oracle's loops and torch's CPU kernels
ATOL, RTOL = 2e-4, 2e-4


def _sd(case, keys_shapes):
    return {k: gu.fill_value(k, s, tag=case) for k, s in keys_shapes.items()}


def _bn(case, prefix, C, mode):
    if mode == 0:
        return None
    return orc.BN(mode, gu.fill_value(f"{prefix}.weight", (C,), case),
                  gu.fill_value(f"{prefix}.bias", (C,), case),
                  gu.fill_value(f"{prefix}.running_mean", (C,), case),
                  gu.fill_value(f"{prefix}.running_var", (C,), case))


def _inputs(c):
    xs = [gu.det_normal(f"{c['name']}.x{i}", tuple(s)) for i, s in enumerate(c["inputs"])]
    if c["name"].endswith("relu_in"):
        xs = [np.maximum(x, 0) for x in xs]
    return xs


def _chk(case, key, arr, scale=1.0):
    """Forward outputs: strict elementwise.  Gradients (scale > 1): tolerance relative to the
    tensor's magnitude, with the ReLU-mask outlier allowance of golden_util._within."""
    if scale == 1.0:
        gu.check(BLK, f"{case}/{key}", arr, ATOL, RTOL, what=f"{case}/{key}")
    else:
        gu.check(BLK, f"{case}/{key}", arr, ATOL, RTOL, what=f"{case}/{key}", scale_rtol=2e-4,
                 outlier_frac=0.05, outlier_rtol=0.02)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["kind"] == "down"])
def test_downsampler(name):
    c = CASES[name]
    ci, co, mode = c["cin"], c["cout"], c["bn_mode"]
    (x,) = _inputs(c)
    w = gu.fill_value("conv.weight", (co - ci, ci, 3, 3), name)
    b = gu.fill_value("conv.bias", (co - ci,), name)
    bn = _bn(name, "bn", co, mode)
    y, pre = orc.downsampler_fwd(x, w, b, bn)
    _chk(name, "y", y)
    gy = gu.det_normal(f"{name}.gy", y.shape)
    dx, dw, db, dg, dbt = orc.downsampler_bwd(x, w, y, pre, gy, bn)
    _chk(name, "dx0", dx, 4)
    _chk(name, "d:conv.weight", dw, 20)
    _chk(name, "d:conv.bias", db, 20)
    if mode:
        _chk(name, "d:bn.weight", dg, 20)
        _chk(name, "d:bn.bias", dbt, 20)
    if mode == 2:
        _chk(name, "buf:bn.running_mean", bn.running_mean)
        _chk(name, "buf:bn.running_var", bn.running_var)


def test_downsampler_odd_size_raises():
    x = np.zeros((1, 3, 15, 24), np.float32)
    with pytest.raises(RuntimeError):
        orc.downsampler_fwd(x, np.zeros((13, 3, 3, 3), np.float32), np.zeros(13, np.float32))


NB_CONVS = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["kind"] == "nb1d"])
def test_nb1d(name):
    c = CASES[name]
    C, d, mode = c["chann"], c["dilated"], c["bn_mode"]
    if name.endswith("bn0_drop"):
        mode = 0
    (x,) = _inputs(c)
    shapes = [(C, C, 3, 1), (C, C, 1, 3), (C, C, 3, 1), (C, C, 1, 3)]
    ws = [gu.fill_value(f"{n}.weight", s, name) for n, s in zip(NB_CONVS, shapes)]
    bs = [gu.fill_value(f"{n}.bias", (C,), name) for n in NB_CONVS]
    bn1, bn2 = _bn(name, "bn1", C, mode), _bn(name, "bn2", C, mode)
    ds = None
    if c.get("drop_scale_key"):
        keep = (gu.det_uniform(c["drop_scale_key"], (x.shape[0], C, 1, 1), 0, 1) >= c["p"])
        ds = (keep.astype(np.float32) / (1.0 - c["p"])).reshape(-1)
    y, saved = orc.nb1d_fwd(x, ws, bs, d, bn1, bn2, ds, save=True)
    _chk(name, "y", y)
    gy = gu.det_normal(f"{name}.gy", y.shape)
    dx, dws, dbs, dgs, dbts = orc.nb1d_bwd(x, ws, d, y, saved, gy, bn1, bn2, ds)
    _chk(name, "dx0", dx, 4)
    for n, dw, db in zip(NB_CONVS, dws, dbs):
        _chk(name, f"d:{n}.weight", dw, 20)
        _chk(name, f"d:{n}.bias", db, 20)
    if mode:
        for i, p in enumerate(("bn1", "bn2")):
            _chk(name, f"d:{p}.weight", dgs[i], 20)
            _chk(name, f"d:{p}.bias", dbts[i], 20)
    if mode == 2:
        _chk(name, "buf:bn1.running_mean", bn1.running_mean)
        _chk(name, "buf:bn2.running_var", bn2.running_var)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["kind"] == "up"])
def test_upsampler(name):
    c = CASES[name]
    ci, co, mode = c["cin"], c["cout"], c["bn_mode"]
    (x,) = _inputs(c)
    w = gu.fill_value("conv.weight", (ci, co, 3, 3), name)
    b = gu.fill_value("conv.bias", (co,), name)
    bn = _bn(name, "bn", co, mode)
    y, pre = orc.upsampler_fwd(x, w, b, bn)
    _chk(name, "y", y)
    gy = gu.det_normal(f"{name}.gy", y.shape)
    dx, dw, db, dg, dbt = orc.upsampler_bwd(x, w, y, pre, gy, bn)
    _chk(name, "dx0", dx, 4)
    _chk(name, "d:conv.weight", dw, 20)
    _chk(name, "d:conv.bias", db, 20)
    if mode:
        _chk(name, "d:bn.weight", dg, 20)
        _chk(name, "d:bn.bias", dbt, 20)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["kind"] == "outconv"])
def test_outconv(name):
    c = CASES[name]
    (x,) = _inputs(c)
    w = gu.fill_value("weight", (16, c["ncls"], 2, 2), name)
    b = gu.fill_value("bias", (c["ncls"],), name)
    y = orc.convT2d_fwd(x, w, b, 2, 0, 0)
    _chk(name, "y", y)
    gy = gu.det_normal(f"{name}.gy", y.shape)
    dx, dw, db = orc.convT2d_bwd(x, w, gy, 2, 0, 0)
    _chk(name, "dx0", dx, 4)
    _chk(name, "d:weight", dw, 20)
    _chk(name, "d:bias", db, 20)


def test_softmaxconv_forward():
    name = "softmaxconv_16_20"
    (x,) = _inputs(CASES[name])
    w = gu.fill_value("convolution.weight", (16, 20, 2, 2), name)
    b = gu.fill_value("convolution.bias", (20,), name)
    y = orc.softmax_c(orc.convT2d_fwd(x, w, b, 2, 0, 0))
    _chk(name, "y", y)


def test_conv1x1():
    name = "conv1x1_128_20"
    (x,) = _inputs(CASES[name])
    w = gu.fill_value("weight", (20, 128, 1, 1), name)
    b = gu.fill_value("bias", (20,), name)
    y = orc.conv2d_fwd(x, w, b)
    _chk(name, "y", y)
    gy = gu.det_normal(f"{name}.gy", y.shape)
    dx, dw, db = orc.conv2d_bwd(x, w, gy)
    _chk(name, "dx0", dx, 4)
    _chk(name, "d:weight", dw, 20)


# ------------------------------------------------------------------------------ whole networks
NETS = gu.load_npz("nets.npz")


def _state(net):
    return {k: gu.fill_value(k, s, tag=net) for k, s in MAN["nets"][net]["state_dict"]}


@pytest.mark.parametrize("net,style,bn", [("eval_erfnet_20", "eval", True),
                                          ("eval_erfnet_nobn_19", "eval", False),
                                          ("net_classification_20", "train", False),
                                          ("net_regression_0", "train", False)])
def test_whole_net_small(net, style, bn):
    x = gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0)
    y = orc.erfnet_forward(x, _state(net), style, use_bn=bn)
    gu.check(NETS, f"{net}/small/y", y, 1e-3, 1e-3)
    if y.shape[1] > 1:   # identical argmax class map wherever the top-2 margin is not a tie
        margin_ok = np.sort(y, axis=1)[:, -1] - np.sort(y, axis=1)[:, -2] > 2e-3
        am = y.argmax(1).astype(np.uint8)
        assert np.array_equal(am[margin_ok], NETS[f"{net}/small/argmax"][margin_ok])


def test_param_counts_match_survey():
    # parameter counts measured on the reference (eval/erfnet.py, train/net_*.py)
    expect = {"eval_erfnet_20": 2066836, "eval_erfnet_nobn_19": 2066642,
              "net_classification_20": 2064405, "net_regression_0": 2063150,
              "imagenet_1000": 2003044}
    for k, v in expect.items():
        assert MAN["nets"][k]["n_params"] == v
