"""GPU parity tests: the HIP path (through the C ABI of liberfnet_hip.so) against
  (1) the golden vectors generated from the reference, block by block and for whole networks,
  (2) the plain-C CPU oracle on the same seeded inputs (with shared saved activations, which makes
      the backward comparison strict), and
  (3) size-independent properties at the full benchmark size.
Tolerances: fp32, 1e-3 absolute on logits (the north-star bound); tighter where stated."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

import golden_util as gu
from test_boundary_cpu import MODELS, first, make_model

pytestmark = pytest.mark.gpu

MAN = gu.load_json("manifest.json")
BLK = gu.load_npz("blocks.npz")
NETS = gu.load_npz("nets.npz")
CASES = {c["name"]: c for c in MAN["block_cases"]}
DEV = "cuda:0"

FWD_ATOL, FWD_RTOL = 2e-4, 2e-4


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def chk_fwd(case, key, arr):
    gu.check(BLK, f"{case}/{key}", arr, FWD_ATOL, FWD_RTOL, what=f"{case}/{key}")


# Gradients against the goldens: a ReLU whose pre-activation is within rounding of zero can open on
# one side and not on the other, which changes the few gradient elements that pixel feeds; 0.5 % of
# the elements may therefore miss the tolerance (by at most 2 % of the tensor's scale).  Two cases
# need more room in the weight gradient of their FIRST convolution, which sums over every pixel of
# three later ReLUs (24 and 65 of 4096 sampled elements on gfx950); the same blocks compare strictly
# in test_nb1d_c_abi_against_oracle / test_nb1d_c_abi_batchnorm_against_oracle, where both sides
# use the same masks.
OUTLIER_FRAC = {"nb1d_128_d16_bn1": 0.01, "nb1d_64_d2_bn0_drop": 0.02}


def chk_grad(case, key, arr):
    gu.check(BLK, f"{case}/{key}", arr, FWD_ATOL, FWD_RTOL, what=f"{case}/{key}", scale_rtol=3e-4,
             outlier_frac=OUTLIER_FRAC.get(case, 0.005), outlier_rtol=0.02)


def build_block(c):
    from erfnet_pytorch_amd import erfnet_blocks as eb
    kind, bn = c["kind"], c["bn_mode"] > 0
    if kind == "down":
        return eb.DownsamplerBlock(c["cin"], c["cout"], use_batchnorm=bn)
    if kind == "up":
        return eb.UpsamplerBlock(c["cin"], c["cout"], use_batchnorm=bn)
    if kind == "nb1d":
        bn = bn and not c["name"].endswith("bn0_drop")
        return eb.non_bottleneck_1d(c["chann"], c.get("p", 0.0), c["dilated"],
                                    use_dropout=bool(c.get("drop_scale_key")), use_batchnorm=bn)
    if kind == "outconv":
        return eb.OutputConvTranspose(c["cin"], c["ncls"])
    if kind == "softmaxconv":
        return eb.SoftMaxConv(c["cin"], c["ncls"], 0.1, False)
    if kind == "ladder":
        return eb.LadderBlock(c["noutput"])
    raise KeyError(kind)


def launches():
    from erfnet_pytorch_amd import _lib
    return _lib.launch_count()


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["kind"] != "conv1x1"])
def test_block_against_golden(name, monkeypatch):
    c = CASES[name]
    mod = gu.fill_state_dict(build_block(c), tag=name).to(DEV)
    mod.train(c["bn_mode"] == 2)
    if c.get("drop_scale_key"):
        # inject the keep mask the fixture was generated with (nn.Dropout2d's RNG is not portable)
        from erfnet_pytorch_amd import erfnet_blocks as eb
        n, ch = c["inputs"][0][0], c["inputs"][0][1]
        keep = (gu.det_uniform(c["drop_scale_key"], (n, ch, 1, 1), 0, 1) >= c["p"]).astype(np.float32)
        scale = t(keep.reshape(n, ch) / (1.0 - c["p"]))
        monkeypatch.setattr(eb, "_dropout2d_scale", lambda x, p, active: scale)
        mod.train(True)
    xs = []
    for i, s in enumerate(c["inputs"]):
        x = gu.det_normal(f"{name}.x{i}", tuple(s))
        if name.endswith("relu_in"):
            x = np.maximum(x, 0)
        xs.append(t(x).requires_grad_(True))
    before = launches()
    y = mod(*xs)
    yv = y.detach().cpu().numpy()               # (the output convolution's logits are written on first use)
    assert launches() > before, "no HIP kernel was launched"
    chk_fwd(name, "y", yv)
    gy = t(gu.det_normal(f"{name}.gy", tuple(y.shape)))
    y.backward(gy)
    for i, x in enumerate(xs):
        chk_grad(name, f"dx{i}", x.grad.cpu().numpy())
    for k, p in mod.named_parameters():
        if f"{name}/d:{k}" in BLK or f"{name}/d:{k}#samp" in BLK:
            assert p.grad is not None, k
            chk_grad(name, f"d:{k}", p.grad.cpu().numpy())
    if c["bn_mode"] == 2:
        for k, b in mod.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                chk_fwd(name, f"buf:{k}", b.cpu().numpy())
            if k.endswith("num_batches_tracked") and (c["kind"] != "nb1d" or build_block(c).use_batchnorm):
                assert int(b) == 1, k       # counted by the statistics kernel, once per forward


def test_conv1x1_against_golden():
    from erfnet_pytorch_amd import functional as hipf
    name = "conv1x1_128_20"
    x = t(gu.det_normal(f"{name}.x0", (2, 128, 6, 10))).requires_grad_(True)
    w = t(gu.fill_value("weight", (20, 128, 1, 1), name)).requires_grad_(True)
    b = t(gu.fill_value("bias", (20,), name)).requires_grad_(True)
    y = hipf.conv2d(x, w, b)
    chk_fwd(name, "y", y.detach().cpu().numpy())
    y.backward(t(gu.det_normal(f"{name}.gy", tuple(y.shape))))
    chk_grad(name, "dx0", x.grad.cpu().numpy())
    chk_grad(name, "d:weight", w.grad.cpu().numpy())
    chk_grad(name, "d:bias", b.grad.cpu().numpy())


def test_downsampler_odd_size_raises():
    from erfnet_pytorch_amd.erfnet_blocks import DownsamplerBlock
    blk = DownsamplerBlock(16, 64).to(DEV)
    with pytest.raises(RuntimeError):
        blk(torch.zeros(1, 16, 15, 24, device=DEV))


# ----------------------------------------------------------------- raw C ABI against the C oracle
def _dptr(x):
    return C.c_void_p(x.data_ptr())


def _nhwc(a):
    """numpy NCHW -> device tensor holding NHWC data."""
    return t(np.ascontiguousarray(a.transpose(0, 2, 3, 1)))


def _to_nchw(x, shape):
    n, c, h, w = shape
    return x.cpu().numpy().reshape(n, h, w, c).transpose(0, 3, 1, 2)


@pytest.mark.parametrize("C_,dil,N,H,W", [(64, 4, 2, 20, 24), (128, 16, 1, 20, 24), (16, 1, 3, 17, 33),
                                          (32, 2, 1, 9, 70)])
def test_nb1d_c_abi_against_oracle(C_, dil, N, H, W):
    """erf_nb1d_forward / erf_nb1d_backward on raw device pointers vs the oracle, the backward fed
    with the oracle's saved activations so that both sides use identical ReLU masks."""
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    tag = f"abi_nb1d_{C_}_{dil}"
    x = np.maximum(gu.det_normal(tag + ".x", (N, C_, H, W)), 0)
    shapes = [(C_, C_, 3, 1), (C_, C_, 1, 3), (C_, C_, 3, 1), (C_, C_, 1, 3)]
    names = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")
    ws = [gu.fill_value(f"{n}.weight", s, tag) for n, s in zip(names, shapes)]
    bs = [gu.fill_value(f"{n}.bias", (C_,), tag) for n in names]
    y_ref, sv = orc.nb1d_fwd(x, ws, bs, dil, save=True)
    gy = gu.det_normal(tag + ".gy", y_ref.shape)
    dx_ref, dws_ref, dbs_ref, _, _ = orc.nb1d_bwd(x, ws, dil, y_ref, sv, gy)

    dx_, dws, dbs = _nhwc(x), [t(w) for w in ws], [t(b) for b in bs]
    y_ref_d, gy_d = _nhwc(y_ref), _nhwc(gy)      # keep the device copies alive across the calls
    nel = x.size
    y = torch.empty(nel, device=DEV)
    save = [torch.empty(nel, device=DEV) for _ in range(5)]
    wsb = torch.empty(int(lib.erf_workspace_bytes(N * H * W, C_)), dtype=torch.uint8, device=DEV)
    arr = lambda ts: (C.c_void_p * len(ts))(*[x_.data_ptr() for x_ in ts])
    _lib.check(lib.erf_nb1d_forward(_dptr(dx_), arr(dws), arr(dbs), None, None, None, dil, _dptr(y),
                                    arr(save), N, C_, H, W, _dptr(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    gu.check_arrays(_to_nchw(y, x.shape), y_ref, FWD_ATOL, FWD_RTOL, what="y")
    for got, ref, nm in ((save[0], sv[0], "t1"), (save[2], sv[2], "t2"), (save[3], sv[3], "t3")):
        gu.check_arrays(_to_nchw(got, x.shape), ref, FWD_ATOL, FWD_RTOL, what=nm)

    # backward with the oracle's activations: masks identical on both sides -> strict comparison
    save_ref = [_nhwc(sv[0]), None, _nhwc(sv[2]), _nhwc(sv[3]), None]
    parr = (C.c_void_p * 5)(*[None if s is None else s.data_ptr() for s in save_ref])
    dx = torch.empty(nel, device=DEV)
    gdw = [torch.empty(w.size, device=DEV) for w in ws]
    gdb = [torch.empty(C_, device=DEV) for _ in range(4)]
    none2 = (C.c_void_p * 2)(None, None)
    _lib.check(lib.erf_nb1d_backward(_dptr(dx_), arr(dws), None, None, None, dil, _dptr(y_ref_d),
                                     parr, _dptr(gy_d), _dptr(dx), arr(gdw), arr(gdb), none2,
                                     none2, N, C_, H, W, 0, _dptr(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    gu.check_arrays(_to_nchw(dx, x.shape), dx_ref, 5e-4, 5e-4, what="dx", scale_rtol=1e-4)
    # the same call with the incoming gradient already masked by (y > 0) and dx masked by (x > 0):
    # parameter gradients bit-identical, dx identical wherever x > 0 and zero elsewhere
    gy_m = (gy_d * (y_ref_d > 0)).contiguous()
    dx2 = torch.empty(nel, device=DEV)
    gdw2 = [torch.empty(w.size, device=DEV) for w in ws]
    gdb2 = [torch.empty(C_, device=DEV) for _ in range(4)]
    _lib.check(lib.erf_nb1d_backward(_dptr(dx_), arr(dws), None, None, None, dil, _dptr(y_ref_d),
                                     parr, _dptr(gy_m), _dptr(dx2), arr(gdw2), arr(gdb2), none2,
                                     none2, N, C_, H, W, _lib.BWD_GY_PREMASKED | _lib.BWD_MASK_DX,
                                     _dptr(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(gdw[i], gdw2[i]) and torch.equal(gdb[i], gdb2[i])
    assert torch.equal(dx2, dx * (dx_.reshape(-1) > 0))
    for i in range(4):
        gu.check_arrays(gdw[i].cpu().numpy().reshape(ws[i].shape), dws_ref[i], 5e-4, 5e-4,
                        what=f"dw{i}", scale_rtol=2e-4)
        gu.check_arrays(gdb[i].cpu().numpy(), dbs_ref[i], 5e-4, 5e-4, what=f"db{i}", scale_rtol=2e-4)


@pytest.mark.parametrize("C_,dil,N,H,W,mode,drop", [(64, 2, 2, 20, 24, 2, False), (128, 4, 2, 12, 20, 2, True),
                                                    (16, 1, 3, 17, 33, 2, False), (64, 8, 1, 20, 24, 1, True)])
def test_nb1d_c_abi_batchnorm_against_oracle(C_, dil, N, H, W, mode, drop):
    """The BatchNorm variants of the block (train/erfnet_blocks.py:35-73) on raw pointers vs the
    oracle: forward incl. batch statistics and running-statistics update, then the backward fed with
    the oracle's saved activations AND saved mean / 1/std, so both sides differentiate the same
    function at the same point and every gradient (dx, dW, db, dgamma, dbeta) compares strictly."""
    from erfnet_pytorch_amd import _lib
    from oracle import oracle as orc
    lib = _lib.lib()
    tag = f"abi_nb1d_bn_{C_}_{dil}_{mode}"
    x = np.maximum(gu.det_normal(tag + ".x", (N, C_, H, W)), 0)
    shapes = [(C_, C_, 3, 1), (C_, C_, 1, 3), (C_, C_, 3, 1), (C_, C_, 1, 3)]
    names = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")
    ws = [gu.fill_value(f"{n}.weight", s, tag) for n, s in zip(names, shapes)]
    bs = [gu.fill_value(f"{n}.bias", (C_,), tag) for n in names]
    bn_np = [dict(gamma=gu.det_uniform(f"{tag}.bn{i}.g", (C_,), 0.5, 1.5), beta=0.1 * gu.det_normal(f"{tag}.bn{i}.b", (C_,)),
                  running_mean=0.1 * gu.det_normal(f"{tag}.bn{i}.rm", (C_,)),
                  running_var=gu.det_uniform(f"{tag}.bn{i}.rv", (C_,), 0.5, 1.5)) for i in (1, 2)]
    obn = [orc.BN(mode, **b) for b in bn_np]
    ds = None
    if drop:
        ds = ((gu.det_uniform(tag + ".keep", (N, C_), 0, 1) >= 0.3) / 0.7).astype(np.float32)
    y_ref, sv = orc.nb1d_fwd(x, ws, bs, dil, obn[0], obn[1], ds, save=True)
    gy = gu.det_normal(tag + ".gy", y_ref.shape)
    dx_ref, dws_ref, dbs_ref, dgs_ref, dbts_ref = orc.nb1d_bwd(x, ws, dil, y_ref, sv, gy, obn[0], obn[1], ds)

    keep = []           # device tensors referenced by raw pointers

    def dev(a):
        keep.append(t(np.asarray(a, np.float32)))
        return keep[-1]

    def bn_struct(b, save_mean=None, save_invstd=None):
        f = {k: dev(v) for k, v in b.items()}
        sm = dev(np.zeros(C_) if save_mean is None else save_mean)
        si = dev(np.zeros(C_) if save_invstd is None else save_invstd)
        st = _lib.BnStruct(mode, f["gamma"].data_ptr(), f["beta"].data_ptr(), f["running_mean"].data_ptr(),
                           f["running_var"].data_ptr(), sm.data_ptr(), si.data_ptr(), 0.1, 1e-3)
        return st, f, sm, si

    arr = lambda ts: (C.c_void_p * len(ts))(*[None if x_ is None else x_.data_ptr() for x_ in ts])
    x_d, dws, dbs = _nhwc(x), [t(w) for w in ws], [t(b) for b in bs]
    ds_d = None if ds is None else dev(ds)
    nel = x.size
    y = torch.empty(nel, device=DEV)
    save = [torch.empty(nel, device=DEV) for _ in range(5)]
    wsb = torch.empty(int(lib.erf_workspace_bytes(N * H * W, C_)), dtype=torch.uint8, device=DEV)
    fwd = [bn_struct(b) for b in bn_np]
    _lib.check(lib.erf_nb1d_forward(_dptr(x_d), arr(dws), arr(dbs), C.byref(fwd[0][0]), C.byref(fwd[1][0]),
                                    None if ds_d is None else _dptr(ds_d), dil, _dptr(y), arr(save), N, C_, H, W,
                                    _dptr(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    gu.check_arrays(_to_nchw(y, x.shape), y_ref, FWD_ATOL, FWD_RTOL, what="y")
    for i in (0, 1, 2, 3, 4):
        gu.check_arrays(_to_nchw(save[i], x.shape), sv[i], FWD_ATOL, FWD_RTOL, what=f"save{i}", scale_rtol=1e-5)
    if mode == 2:
        for (st, f, sm, si), o in zip(fwd, obn):
            gu.check_arrays(sm.cpu().numpy(), o.save_mean, 1e-5, 1e-5, what="batch mean")
            gu.check_arrays(si.cpu().numpy(), o.save_invstd, 1e-5, 1e-4, what="batch 1/std")
            gu.check_arrays(f["running_mean"].cpu().numpy(), o.running_mean, 1e-6, 1e-5, what="running_mean")
            gu.check_arrays(f["running_var"].cpu().numpy(), o.running_var, 1e-6, 1e-5, what="running_var")

    # backward at the oracle's point: its activations, its batch statistics
    bwd = [bn_struct(b, o.save_mean, o.save_invstd) for b, o in zip(bn_np, obn)]
    save_ref = [_nhwc(sv[i]) for i in range(5)]
    y_ref_d, gy_d = _nhwc(y_ref), _nhwc(gy)
    dx = torch.empty(nel, device=DEV)
    gdw = [torch.empty(w.size, device=DEV) for w in ws]
    gdb = [torch.empty(C_, device=DEV) for _ in range(4)]
    gdg = [torch.empty(C_, device=DEV) for _ in range(2)]
    gdbt = [torch.empty(C_, device=DEV) for _ in range(2)]
    _lib.check(lib.erf_nb1d_backward(_dptr(x_d), arr(dws), C.byref(bwd[0][0]), C.byref(bwd[1][0]),
                                     None if ds_d is None else _dptr(ds_d), dil, _dptr(y_ref_d), arr(save_ref),
                                     _dptr(gy_d), _dptr(dx), arr(gdw), arr(gdb), arr(gdg), arr(gdbt), N, C_, H, W, 0,
                                     _dptr(wsb), wsb.numel(), None))
    torch.cuda.synchronize()
    gu.check_arrays(_to_nchw(dx, x.shape), dx_ref, 5e-4, 5e-4, what="dx", scale_rtol=2e-4)
    for i in range(4):
        gu.check_arrays(gdw[i].cpu().numpy().reshape(ws[i].shape), dws_ref[i], 5e-4, 5e-4, what=f"dw{i}",
                        scale_rtol=2e-4)
        # a bias in front of a train-mode BatchNorm has gradient zero up to rounding of large sums
        gu.check_arrays(gdb[i].cpu().numpy(), dbs_ref[i], 5e-4, 5e-4, what=f"db{i}", scale_rtol=2e-4)
    for i in range(2):
        gu.check_arrays(gdg[i].cpu().numpy(), dgs_ref[i], 5e-4, 5e-4, what=f"dgamma{i}", scale_rtol=2e-4)
        gu.check_arrays(gdbt[i].cpu().numpy(), dbts_ref[i], 5e-4, 5e-4, what=f"dbeta{i}", scale_rtol=2e-4)


def test_c_abi_rejects_bad_arguments():
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    x = torch.zeros(16, device=DEV)
    # odd height: rejected before any launch, message available
    rc = lib.erf_downsampler_forward(_dptr(x), 0, 0, 0, 0, _dptr(x), _dptr(x), None, _dptr(x), None,
                                     1, 16, 15, 24, 64, None, 0, None)
    assert rc != 0 and b"even" in lib.erf_last_error()
    # workspace too small
    rc = lib.erf_nb1d_forward(_dptr(x), (C.c_void_p * 4)(), (C.c_void_p * 4)(), None, None, None, 1,
                              _dptr(x), None, 1, 16, 8, 8, None, 0, None)
    assert rc != 0 and b"workspace" in lib.erf_last_error()


# --------------------------------------------------------------------------------- whole networks
@pytest.mark.parametrize("key", [k for k in MODELS if k != "imagenet_1000"])
def test_whole_net_small(key):
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    x = t(gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0))
    with torch.no_grad():
        y = first(net(x)).cpu().numpy()
    gu.check(NETS, f"{key}/small/y", y, 1e-3, 1e-3)
    if y.shape[1] > 1:
        srt = np.sort(y, axis=1)
        ok = (srt[:, -1] - srt[:, -2]) > 2e-3
        assert np.array_equal(y.argmax(1).astype(np.uint8)[ok], NETS[f"{key}/small/argmax"][ok])


def test_only_encode_gpu():
    key = "eval_erfnet_20"
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    x = t(gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0))
    with torch.no_grad():
        y = net(x, only_encode=True).cpu().numpy()
    gu.check(NETS, "eval_erfnet_20/small/y_encode", y, 1e-3, 1e-3)


@pytest.mark.parametrize("key", ["eval_erfnet_20", "net_classification_20"])
def test_whole_net_512x1024_argmax(key):
    """Cityscapes-sized input: logits within 1e-3 of the reference and identical argmax class map
    (pixels whose top-2 margin is below the tolerance are ties and excluded)."""
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    x = t(gu.det_uniform("net.x_big", (1, 3, 512, 1024), 0.0, 1.0))
    with torch.no_grad():
        y = first(net(x)).cpu().numpy()
    gu.check(NETS, f"{key}/big/y", y, 1e-3, 1e-3)
    np.testing.assert_allclose(y[:, :, 224:288, 480:544], NETS[f"{key}/big/crop"], atol=1e-3, rtol=1e-3)
    ok = NETS[f"{key}/big/margin"].astype(np.float32) > 2.5e-3
    assert ok.mean() > 0.9
    assert np.array_equal(y.argmax(1).astype(np.uint8)[ok], NETS[f"{key}/big/argmax"][ok])


def test_imagenet_encoder():
    key = "imagenet_1000"
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    x = t(gu.det_uniform("net.x_imagenet", (2, 3, 224, 224), 0.0, 1.0))
    with torch.no_grad():
        y = net(x).cpu().numpy()
    gu.check(NETS, "imagenet_1000/y", y, 1e-3, 1e-3)


def test_whole_net_bn_train_forward_backward():
    """eval/erfnet.ERFNet(20) in train mode (batch statistics, dropout probability 0):
    logits, loss, every parameter gradient and the updated running statistics."""
    key = "eval_erfnet_20_train"
    net = gu.fill_state_dict(make_model("eval_erfnet_20"), tag=key).to(DEV)
    for sub in net.modules():
        if isinstance(sub, torch.nn.Dropout2d):
            sub.p = 0
    net.train()
    x = t(gu.det_uniform("net.x_train", (2, 3, 64, 128), 0.0, 1.0))
    lab = t(gu.det_labels("net.lab_train", (2, 64, 128), -1, 20))
    out = net(x)
    loss = torch.nn.CrossEntropyLoss(ignore_index=-1)(out, lab)
    loss.backward()
    gu.check(NETS, f"{key}/y", out.detach().cpu().numpy(), 2e-3, 2e-3)
    assert abs(loss.item() - float(NETS[f"{key}/loss"][0])) < 1e-3
    # Gradients of this configuration (random weights, batch statistics over as few as 256 samples
    # per channel, 23 stacked BN blocks) are ill-conditioned in fp32: the reference's own modules in
    # fp32 differ from the same modules in fp64 by 2-3e-2 relative L2 on every parameter (measured
    # on the GPU box with scripts/diag_bn_train.py), so that is the resolution of this check.
    # Parameters whose gradient is analytically zero (conv biases feeding a BatchNorm) hold noise
    # only and are skipped.  Tight whole-network gradient parity is pinned by the BN-free
    # two-Adam-step test below.
    checked = 0
    for k, p in net.named_parameters():
        name = f"{key}/d:{k}"
        if name in NETS:
            ref = NETS[name].astype(np.float64)
            got = p.grad.cpu().numpy().astype(np.float64)
        elif name + "#samp" in NETS:
            ref = NETS[name + "#samp"].astype(np.float64)
            flat = p.grad.cpu().numpy().reshape(-1)
            got = flat[gu.sample_index(name, flat.size)].astype(np.float64)
        else:
            continue
        if np.linalg.norm(ref) < 1e-6 * max(1.0, ref.size ** 0.5):
            continue
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 0.1, (k, rel)
        checked += 1
    assert checked > 150
    for k, b in net.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            gu.check(NETS, f"{key}/buf:{k}", b.cpu().numpy(), 1e-4, 1e-3, what=k)


def test_two_adam_steps_match_reference():
    """The reference's training step (train/main.py:43-51,213,297-392): CrossEntropy(ignore -1),
    Adam(lr 1e-4, eps 1e-4, wd 1e-6); losses and parameter checksums after two steps."""
    ref = gu.load_json("trainstep.json")
    net = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    crit = CrossEntropyLoss2d()
    x = t(gu.det_uniform("trainstep.x", (2, 3, 64, 128), 0.0, 1.0))
    lab = t(gu.det_labels("trainstep.lab", (2, 64, 128), -1, 20))
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


def test_direct_gradient_write_matches_autograd_accumulation():
    """GradientReducer(direct_write=True) lets the kernels write each weight gradient into its
    bucket slice; the values must equal what autograd's accumulation leaves there, step after
    step, and the reference's two-step training numbers must still come out."""
    from erfnet_pytorch_amd.ddp import GradientReducer
    ref = gu.load_json("trainstep.json")
    x = t(gu.det_uniform("trainstep.x", (2, 3, 64, 128), 0.0, 1.0))
    lab = t(gu.det_labels("trainstep.lab", (2, 64, 128), -1, 20))
    crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
    grads = {}
    for direct in (False, True):
        net = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV).train()
        red = GradientReducer(net, direct_write=direct)
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], 1e-4, (0.9, 0.999),
                               eps=1e-4, weight_decay=1e-6)
        per_step = []
        for step in range(3):
            red.zero_grad()
            out, _ = net(x)
            loss = crit(out, lab)
            loss.backward()
            red.finish()
            per_step.append({k: p.grad.detach().clone() for k, p in net.named_parameters()
                             if p.grad is not None})
            opt.step()
            if step < 2:
                assert abs(loss.item() - ref["losses"][step]) < 1e-3
        grads[direct] = per_step
        red.remove()
    for step in range(3):
        a, b = grads[False][step], grads[True][step]
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (step, k)


@pytest.mark.parametrize("shape,k", [((2, 64, 128), 20), ((1, 7, 13), 1), ((3, 33, 17), 5), ((1, 9, 11), 32),
                                     ((1, 9, 11), 40)])
def test_cross_entropy_against_torch_and_oracle(shape, k):
    """CrossEntropyLoss2d (train/evaluation_functions.py:6-14): loss and gradient of the HIP kernels
    against the oracle's float64 restatement and torch's own op on the same tensors."""
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    from oracle import oracle as orc
    n, h, w = shape
    x_np = (gu.det_normal(f"ce.x.{k}", (n, k, h, w)) * 3.0).astype(np.float32)
    lab_np = gu.det_labels(f"ce.lab.{k}", (n, h, w), -1, k)
    x = t(x_np).requires_grad_(True)
    lab = t(lab_np)
    n0 = launches()
    loss = CrossEntropyLoss2d()(x, lab)
    (loss * 1.7).backward()
    assert launches() > n0
    ref_loss, ref_g = orc._ce_loss_and_grad(x_np, lab_np)
    assert abs(loss.item() - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    np.testing.assert_allclose(x.grad.cpu().numpy(), 1.7 * ref_g, rtol=2e-5, atol=1e-9)
    x2 = t(x_np).requires_grad_(True)
    l2 = torch.nn.functional.cross_entropy(x2, lab, ignore_index=-1)
    (l2 * 1.7).backward()
    assert abs(loss.item() - l2.item()) <= 1e-5 * max(1.0, abs(l2.item()))
    assert torch.allclose(x.grad, x2.grad, rtol=2e-5, atol=1e-9)


def test_cross_entropy_nothing_labelled_is_nan():
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    x = t(gu.det_normal("ce.x.nan", (1, 4, 8, 8)))
    lab = torch.full((1, 8, 8), -1, dtype=torch.int64, device=DEV)
    assert torch.isnan(CrossEntropyLoss2d()(x, lab))
    assert torch.isnan(torch.nn.functional.cross_entropy(x, lab, ignore_index=-1))


@pytest.mark.parametrize("key", ["net_classification_20", "net_classification_ladder_20", "net_regression_ladder_0",
                                 "net_regression_ladder_small_0"])
def test_relu_mask_hand_over_leaves_gradients_unchanged(key, monkeypatch):
    """Neighbouring blocks hand the ReLU mask of the gradient over (consumer masks its dx, producer
    skips its masking pass).  Every parameter gradient and the input gradient must be bit-identical
    to the run in which no block knows its neighbour — also in the ladder nets, whose encoder
    outputs have two consumers (the next block and a LadderBlock), and also for a second backward
    through a retained graph, when the tokens of the first pass have been used up."""
    from erfnet_pytorch_amd import functional as hipf
    x_np = gu.det_uniform("handover.x", (2, 3, 64, 128), 0.0, 1.0)
    g_np = None
    res = []
    for hand_over in (True, False):
        if not hand_over:
            monkeypatch.setattr(hipf, "_input_token", lambda x: None)
        net = gu.fill_state_dict(make_model(key), tag="handover." + key).to(DEV).train()
        for m in net.modules():           # dropout off: both runs must see the same network
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        x = t(x_np).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        n0 = launches()
        out = first(net(x))
        if g_np is None:
            g_np = gu.det_normal("handover.g", tuple(out.shape))
        g = t(g_np)
        out.backward(g, retain_graph=True)
        one = ({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, x.grad.clone(),
               launches() - n0)
        net.zero_grad(set_to_none=True)
        x.grad = None
        out.backward(g)
        two = ({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, x.grad.clone())
        res.append((one, two))
    ((ga, xa, la), (ga2, xa2)), ((gb, xb, lb), (gb2, xb2)) = res
    assert la < lb                        # the masking passes really were skipped
    assert ga.keys() == gb.keys() == ga2.keys() == gb2.keys()
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
        assert torch.equal(ga2[k], gb[k]), ("second backward", k)
        assert torch.equal(gb2[k], gb[k]), ("second backward, no hand-over", k)
    assert torch.equal(xa, xb) and torch.equal(xa2, xb) and torch.equal(xb2, xb)


def _loss_head_run(mode, dtype=torch.float32, ncls=20, shape=(2, 16, 24, 40)):
    """Output convolution + loss on seeded inputs; mode selects how the logits are consumed."""
    from erfnet_pytorch_amd import functional as hipf
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d, MSELossWeighted
    n, c, h, w = shape
    x = t(np.maximum(gu.det_normal("losshead.x", shape), 0)).to(dtype).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    wt = t(0.3 * gu.det_normal("losshead.w", (16, ncls, 2, 2))).requires_grad_(True)
    b = t(0.1 * gu.det_normal("losshead.b", (ncls,))).requires_grad_(True)
    lab = t(gu.det_labels("losshead.lab", (n, 2 * h, 2 * w), -1, ncls))
    other = t(gu.det_normal("losshead.other", (n, ncls, 2 * h, 2 * w)))
    n0 = launches()
    y = hipf.outconv(x, wt, b)
    loss = CrossEntropyLoss2d()(y, lab)
    if mode == "scaled":
        loss = loss * 0.37
    elif mode == "two_consumers":          # mean-teacher: the same logits feed the consistency loss
        loss = loss + MSELossWeighted(2.0)(y, other, 0.5)
    elif mode == "two_losses":
        loss = loss + 0.5 * CrossEntropyLoss2d()(y, lab.flip(0))
    loss.backward(retain_graph=(mode == "twice"))
    if mode == "twice":
        first = [v.grad.clone() for v in (x, wt, b)]
        for v in (x, wt, b):
            v.grad = None
        loss.backward()
        for u, v in zip(first, (x, wt, b)):
            assert torch.equal(u, v.grad)
    return float(loss.detach()), [v.grad.float().cpu().numpy() for v in (x, wt, b)], launches() - n0


@pytest.mark.parametrize("mode", ["plain", "scaled", "two_consumers", "two_losses", "twice"])
@pytest.mark.parametrize("dtype,ncls", [(torch.float32, 20), (torch.float32, 7), (torch.float32, 29),
                                        (torch.bfloat16, 20)])
def test_loss_hand_over_matches_separate_kernels(mode, dtype, ncls):
    """CrossEntropyLoss2d's backward fused into the output convolution's (erf_outconv_ce_backward, the
    gradient of the logits never written) against the separate kernels: same loss, gradients equal
    up to the rounding of a recomputed logit — also when the logits have a second consumer, two
    losses, an upstream scale, or the graph is walked twice."""
    from erfnet_pytorch_amd import functional as hipf
    fused = _loss_head_run(mode, dtype, ncls)
    old = hipf.set_loss_handover(False)
    try:
        plain = _loss_head_run(mode, dtype, ncls)
    finally:
        hipf.set_loss_handover(old)
    # (the deferred logits evaluate the loss in erf_outconv_ce_forward, whose block sums group the pixels
    # differently from erf_cross_entropy_forward's: equal to a few fp32 roundings)
    assert abs(fused[0] - plain[0]) <= 2e-6 * abs(plain[0])
    if mode in ("plain", "scaled"):
        assert fused[2] < plain[2]          # fewer launches: the fused kernel really ran
    tol = 2e-5 if dtype == torch.float32 else 1e-2      # bf16: dx is rounded to bf16 on both sides
    for nm, u, v in zip(("dx", "dw", "db"), fused[1], plain[1]):
        gu.check_arrays(u, v, 1e-7, 1e-5, what=f"{mode}/{nm}", scale_rtol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ncls,n,h,w", [(20, 2, 24, 40), (7, 1, 9, 13), (32, 3, 16, 16), (1, 2, 8, 8)])
def test_outconv_ce_forward_against_the_two_kernels(dtype, ncls, n, h, w):
    """erf_outconv_ce_forward (output convolution + CrossEntropyLoss + accuracy counts, the logits never
    stored; train/main.py:327,333,343-347) through the raw C ABI against erf_outconv_forward followed by
    erf_cross_entropy_forward_counts: the log-sum-exp bit for bit, the counts exactly, loss and pixel
    count to fp32 rounding; and the loss against torch's own on the stored logits."""
    import ctypes as C
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    p = lambda q: None if q is None else C.c_void_p(q.data_ptr())
    sfx = "_bf16" if dtype == torch.bfloat16 else ""
    x = t(np.maximum(gu.det_normal("ocef.x", (n, 16, h, w)), 0)).to(dtype).contiguous(memory_format=torch.channels_last)
    wt = t(0.3 * gu.det_normal("ocef.w", (16, ncls, 2, 2)))
    b = t(0.1 * gu.det_normal("ocef.b", (ncls,)))
    lab = t(gu.det_labels("ocef.lab", (n, 2 * h, 2 * w), -1, ncls))
    if ncls > 2:
        lab[0, 0, :4] = ncls + 3            # labels outside [0, K) that are not ignore_index count as unlabelled
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wsb = torch.empty(int(lib.erf_workspace_bytes(n * 4 * h * w, 2 * ncls + 2)), dtype=torch.uint8, device=DEV)
    logits = torch.empty(n, ncls, 2 * h, 2 * w, device=DEV)
    if dtype == torch.bfloat16:
        _lib.check(lib.erf_outconv_forward_bf16(p(x), p(wt), p(b), p(logits), 0, n, 16, h, w, ncls, 0, st))
    else:
        _lib.check(lib.erf_outconv_forward(p(x), p(wt), p(b), p(logits), n, 16, h, w, ncls, 0, st))
    s0, l0, c0 = torch.zeros(2, device=DEV), torch.zeros(n, 2 * h, 2 * w, device=DEV), torch.zeros(2, ncls, dtype=torch.int64, device=DEV)
    _lib.check(lib.erf_cross_entropy_forward_counts(p(logits), p(lab), p(s0), p(l0), p(c0), n, ncls, 2 * h, 2 * w, -1,
                                                    p(wsb), wsb.numel(), st))
    s1, l1, c1 = torch.zeros(2, device=DEV), torch.zeros(n, 2 * h, 2 * w, device=DEV), torch.zeros(2, ncls, dtype=torch.int64, device=DEV)
    before = launches()
    _lib.check(getattr(lib, "erf_outconv_ce_forward" + sfx)(p(x), p(wt), p(b), p(lab), p(s1), p(l1), p(c1), n, 16, h, w,
                                                           ncls, -1, p(wsb), wsb.numel(), st))
    torch.cuda.synchronize()
    assert launches() == before + 3      # weight pack, the fused kernel, one reduction of loss and counts
    assert torch.equal(l0, l1)
    assert torch.equal(c0, c1)
    assert float(s0[1]) == float(s1[1]) > 0
    assert abs(float(s0[0]) - float(s1[0])) <= 2e-6 * abs(float(s0[0]))
    valid = (lab >= 0) & (lab < ncls)
    ref = torch.nn.functional.cross_entropy(logits.double(), torch.where(valid, lab, torch.full_like(lab, -1)),
                                            ignore_index=-1)
    assert abs(float(ref) - float(s1[0])) <= 1e-5 * max(1.0, abs(float(ref)))
    # without counts
    s2, l2 = torch.zeros(2, device=DEV), torch.zeros(n, 2 * h, 2 * w, device=DEV)
    _lib.check(getattr(lib, "erf_outconv_ce_forward" + sfx)(p(x), p(wt), p(b), p(lab), p(s2), p(l2), None, n, 16, h, w,
                                                           ncls, -1, p(wsb), wsb.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(s1, s2) and torch.equal(l1, l2)


def test_deferred_logits_are_ordinary_logits_to_everybody_but_the_loss():
    """While autograd records, the output convolution defers its kernel (functional._LazyLogits): the
    loss never needs the logits in memory, anything else gets them on first touch — the values of
    THAT forward pass, even when the parameters have moved since."""
    from erfnet_pytorch_amd import functional as hipf
    from erfnet_pytorch_amd.evaluation_functions import ClassificationAccuracy, CrossEntropyLoss2d
    shape, ncls = (2, 16, 12, 20), 20
    x = t(np.maximum(gu.det_normal("losshead.x", shape), 0)).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    wt = t(0.3 * gu.det_normal("losshead.w", (16, ncls, 2, 2))).requires_grad_(True)
    b = t(0.1 * gu.det_normal("losshead.b", (ncls,))).requires_grad_(True)
    lab = t(gu.det_labels("losshead.lab", (2, 24, 40), -1, ncls))
    with torch.no_grad():
        eager = hipf.outconv(x, wt, b).clone()
    # 1. loss only: no logits kernel at all
    n0 = launches()
    y = hipf.outconv(x, wt, b)
    assert type(y) is hipf._LazyLogits and tuple(y.shape) == (2, ncls, 24, 40) and y.requires_grad
    loss = CrossEntropyLoss2d()(y, lab)
    acc = ClassificationAccuracy()(y, lab)           # served from the counts the loss left behind
    loss.backward()
    n_lazy = launches() - n0
    assert hipf._lazy_pending(y) is not None         # still never written
    ref_acc = float(((eager.argmax(1) == lab) & (lab >= 0)).sum()) / float((lab >= 0).sum())
    assert abs(float(acc) - ref_acc) < 1e-6
    got = [v.grad.clone() for v in (x, wt, b)]
    # 2. the parameters move, then somebody looks at the logits of that forward pass
    keep = (wt.detach().clone(), b.detach().clone())
    with torch.no_grad():
        wt.mul_(0.5)
        b.add_(1.0)
    assert torch.equal(y.detach(), eager)
    assert hipf._lazy_pending(y) is None
    assert torch.equal(y.max(1)[1], eager.max(1)[1])
    with torch.no_grad():
        wt.copy_(keep[0])
        b.copy_(keep[1])
    # 3. the same step with eager logits: same gradients, more launches
    for v in (x, wt, b):
        v.grad = None
    old = hipf.set_lazy_logits(False)
    try:
        n0 = launches()
        y2 = hipf.outconv(x, wt, b)
        assert type(y2) is torch.Tensor
        loss2 = CrossEntropyLoss2d()(y2, lab)
        ClassificationAccuracy()(y2, lab)
        loss2.backward()
        n_eager = launches() - n0
    finally:
        hipf.set_lazy_logits(old)
    assert n_lazy <= n_eager
    assert abs(float(loss) - float(loss2)) <= 2e-6 * abs(float(loss2))
    for nm, u, v in zip(("dx", "dw", "db"), got, (x, wt, b)):
        assert torch.equal(u, v.grad), nm
    # 4. a second consumer of the logits (the consistency loss of the mean-teacher loop) materialises them
    for v in (x, wt, b):
        v.grad = None
    y3 = hipf.outconv(x, wt, b)
    total = CrossEntropyLoss2d()(y3, lab) + 0.1 * (y3 * y3).mean()
    total.backward()
    y4 = eager.clone().requires_grad_(True)
    ref_total = torch.nn.functional.cross_entropy(y4, lab, ignore_index=-1) + 0.1 * (y4 * y4).mean()
    assert abs(float(total) - float(ref_total)) <= 1e-5 * abs(float(ref_total))


def test_gradient_of_the_logits_themselves_is_refused_loudly_under_hand_over():
    """torch.autograd.grad(loss, logits) cannot be served by the hand-over (the loss gradient goes
    straight to the output convolution): the pass must raise instead of returning the zero stand-in,
    and must leave nothing behind — a following ordinary backward gives the ordinary gradients.  With
    the hand-over off the gradient of the logits is the real one."""
    from erfnet_pytorch_amd import functional as hipf
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    shape, ncls = (2, 16, 12, 20), 20
    x = t(np.maximum(gu.det_normal("losshead.x", shape), 0)).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    wt = t(0.3 * gu.det_normal("losshead.w", (16, ncls, 2, 2))).requires_grad_(True)
    b = t(0.1 * gu.det_normal("losshead.b", (ncls,))).requires_grad_(True)
    lab = t(gu.det_labels("losshead.lab", (2, 24, 40), -1, ncls))
    y = hipf.outconv(x, wt, b)
    loss = CrossEntropyLoss2d()(y, lab)
    with pytest.raises(RuntimeError, match="set_loss_handover"):
        torch.autograd.grad(loss, y, retain_graph=True)
    loss.backward()
    got = [v.grad.clone() for v in (x, wt, b)]
    for v in (x, wt, b):
        v.grad = None
    old = hipf.set_loss_handover(False)
    try:
        y2 = hipf.outconv(x, wt, b)
        loss2 = CrossEntropyLoss2d()(y2, lab)
        (gl,) = torch.autograd.grad(loss2, y2, retain_graph=True)
        assert float(gl.abs().max()) > 0
        ref = torch.nn.functional.cross_entropy(y2.detach().clone().requires_grad_(True), lab, ignore_index=-1)
        loss2.backward()
    finally:
        hipf.set_loss_handover(old)
    assert abs(float(ref) - float(loss2)) < 1e-5
    for nm, u, v in zip(("dx", "dw", "db"), got, (x, wt, b)):
        gu.check_arrays(u.cpu().numpy(), v.grad.cpu().numpy(), 1e-7, 1e-5, what=nm, scale_rtol=2e-5)


def test_graphed_forward_replays_the_eager_result():
    """A forward captured into a HIP graph (graphs.GraphedForward) returns bit-identical logits on
    replay, for the captured input and for new data of the same shape."""
    from erfnet_pytorch_amd.graphs import GraphedForward
    net = gu.fill_state_dict(make_model("eval_erfnet_nobn_19"), tag="eval_erfnet_nobn_19").to(DEV).eval()
    x1 = t(gu.det_uniform("graph.x1", (1, 3, 64, 128), 0.0, 1.0))
    x2 = t(gu.det_uniform("graph.x2", (1, 3, 64, 128), 0.0, 1.0))
    with torch.no_grad():
        y1, y2 = first(net(x1)).clone(), first(net(x2)).clone()
    g = GraphedForward(net, x1)
    assert torch.equal(first(g(x1)), y1)
    assert torch.equal(first(g(x2)), y2)
    assert torch.equal(first(g(x1)), y1)
    with pytest.raises(ValueError):
        g(x1[:, :, :32])


@pytest.mark.parametrize("C_,d,N,H,W", [(128, 2, 2, 40, 24), (128, 16, 1, 64, 20), (64, 1, 2, 72, 20),
                                        (64, 2, 1, 128, 12), (128, 8, 3, 28, 28), (64, 1, 2, 56, 56),
                                        (128, 1, 1, 17, 33), (16, 1, 2, 40, 48), (16, 1, 1, 17, 33),
                                        (16, 2, 1, 24, 100), (16, 4, 1, 33, 70)])
@pytest.mark.parametrize("bn_mode", [0, 1, 2])
def test_fused_conv_pairs_bit_identical(C_, d, N, H, W, bn_mode):
    """Two chained 1-D convolutions in one launch (pair_f32_kernel, image columns as block tiles;
    pair16_kernel, 2-D tiles with a halo, at 16 channels)
    must reproduce the one-launch-per-convolution path bit for bit: inference output, training
    output, input gradient and every parameter gradient, with and without BatchNorm."""
    from erfnet_pytorch_amd import functional as hipf, _lib
    g = torch.Generator(device="cpu").manual_seed(C_ + d + H)
    x = torch.randn(N, C_, H, W, generator=g).clamp_(min=0).to(DEV).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(C_, C_, *k, generator=g) * 0.05).to(DEV).requires_grad_(True) for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
    bs = [(torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True) for _ in range(4)]
    gy = torch.randn(N, C_, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    bnp = [[(torch.rand(C_, generator=g) + 0.5).to(DEV).requires_grad_(True),
            (torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True)] for _ in range(2)]
    params = ws + bs + (bnp[0] + bnp[1] if bn_mode else [])

    def run(train):
        stats = [(torch.randn(C_, generator=torch.Generator().manual_seed(5 + i)) * 0.1).to(DEV) for i in range(2)]
        var = [(torch.rand(C_, generator=torch.Generator().manual_seed(9 + i)) + 0.5).to(DEV) for i in range(2)]
        kw = {}
        if bn_mode:
            kw = dict(bn1=(bnp[0][0], bnp[0][1], stats[0], var[0]), bn2=(bnp[1][0], bnp[1][1], stats[1], var[1]),
                      bn_mode=bn_mode)
        if not train:
            with torch.no_grad():
                return [hipf.nb1d(x, ws, bs, d, **kw)]
        for q in params:
            q.grad = None
        xx = x.clone().requires_grad_(True)
        y = hipf.nb1d(xx, ws, bs, d, **kw)
        y.backward(gy)
        return [y.detach(), xx.grad] + [q.grad.clone() for q in params] + (stats + var if bn_mode else [])

    res = {}
    prev = _lib.set_option(_lib.OPT_PAIR_FUSION, 1)
    prev16 = _lib.set_option(_lib.OPT_PAIR16, 1)        # (off by default: not faster yet)
    prev_split = _lib.set_option(_lib.OPT_SPLIT_BF16, 0)    # (the split-bf16 path runs every convolution in a launch of its own)
    try:
        for on in (0, 1):
            _lib.set_option(_lib.OPT_PAIR_FUSION, on)
            n0 = launches(); a = run(False); n1 = launches(); b = run(True); n2 = launches()
            res[on] = (a + b, n1 - n0, n2 - n1)
    finally:
        _lib.set_option(_lib.OPT_PAIR_FUSION, prev)
        _lib.set_option(_lib.OPT_PAIR16, prev16)
        _lib.set_option(_lib.OPT_SPLIT_BF16, prev_split)
    # without BatchNorm the forward pair and both backward pairs fuse; inference folds eval-mode
    # BatchNorm into the epilogue (pair possible); batch statistics, or a BatchNorm whose input is
    # kept for backward, sit between the two convolutions of the forward pair, never between those
    # of the backward pairs
    assert res[1][1] == res[0][1] - (0 if bn_mode == 2 else 1)
    assert res[1][2] == res[0][2] - (2 if bn_mode else 3)
    for i, (u, v) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.equal(u, v), f"tensor {i}"


@pytest.mark.parametrize("C_,N,H,W,nblocks", [(128, 1, 64, 24, 3), (64, 2, 72, 20, 2), (16, 1, 16, 32, 2),
                                              (128, 1, 40, 16, 10)])
@pytest.mark.parametrize("bn_mode,dropout", [(0, False), (2, False), (1, True), (0, True)])
def test_nb1d_chain_equals_blockwise_calls(C_, N, H, W, nblocks, bn_mode, dropout):
    """functional.nb1d_chain (erf_nb1d_chain_forward + per-block backward) must give bit-identical
    outputs, input gradient, parameter gradients and running statistics to calling the blocks one
    after the other — including runs longer than the library's segment length, BatchNorm in both
    modes and Dropout2d factors (which keep a block's tail out of the shared launch)."""
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator(device="cpu").manual_seed(C_ * 7 + nblocks)
    x = torch.randn(N, C_, H, W, generator=g).clamp_(min=0).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, C_, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dils = [1, 2, 4, 8, 16, 1, 2, 3, 1, 5][:nblocks]
    blocks = []
    for i in range(nblocks):
        ws = [(torch.randn(C_, C_, *k, generator=g) * (0.5 / C_ ** 0.5)).to(DEV).requires_grad_(True)
              for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
        bs = [(torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True) for _ in range(4)]
        b = dict(weights=ws, biases=bs, dilated=dils[i])
        if bn_mode:
            b["bn_mode"] = bn_mode
            for key in ("bn1", "bn2"):
                b[key] = ((torch.rand(C_, generator=g) + 0.5).to(DEV).requires_grad_(True),
                          (torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True),
                          (torch.randn(C_, generator=g) * 0.1).to(DEV), (torch.rand(C_, generator=g) + 0.5).to(DEV))
        if dropout and i % 2 == 0:
            b["drop_scale"] = (torch.rand(N, C_, generator=g) > 0.3).float().div(0.7).to(DEV)
        blocks.append(b)

    def leaves():
        out = []
        for b in blocks:
            out += b["weights"] + b["biases"]
            if bn_mode:
                out += [b["bn1"][0], b["bn1"][1], b["bn2"][0], b["bn2"][1]]
        return out

    def stats():
        return [t for b in blocks if bn_mode for key in ("bn1", "bn2") for t in b[key][2:]]

    stats0 = [t.clone() for t in stats()]
    res = []
    for chained in (False, True):
        for t, t0 in zip(stats(), stats0):
            t.copy_(t0)
        for q in leaves():
            q.grad = None
        xx = x.clone().requires_grad_(True)
        n0 = launches()
        if chained:
            y = hipf.nb1d_chain(xx, blocks)
        else:
            y = xx
            for b in blocks:
                kw = dict(b)
                y = hipf.nb1d(y, kw.pop("weights"), kw.pop("biases"), kw.pop("dilated"), **kw)
        y.backward(gy)
        got = [y.detach(), xx.grad] + [q.grad.clone() for q in leaves()] + [t.clone() for t in stats()]
        with torch.no_grad():
            yi = hipf.nb1d_chain(x, blocks) if chained else None
        res.append((got, launches() - n0, yi))
    for i, (u, v) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.equal(u, v), f"tensor {i}"
    # inference under no_grad: the training forward's values (eval-mode BatchNorm is folded into
    # the convolution epilogue there, which rounds differently)
    if bn_mode == 0:
        assert torch.equal(res[1][2], res[0][0][0])
    elif bn_mode == 1:
        assert torch.allclose(res[1][2], res[0][0][0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("C_,N,H,W,axis,d1,d2", [(128, 2, 64, 40, 1, 1, 2), (128, 1, 40, 128, 0, 2, 2),
                                                 (64, 1, 128, 24, 1, 4, 1), (64, 2, 20, 200, 0, 1, 3)])
def test_conv_pair_c_abi_against_torch(C_, N, H, W, axis, d1, d2):
    """erf_conv_pair_pack / erf_conv_pair_forward (both line orientations, stored and unstored
    intermediate, residual) against stock torch fp32 convolutions: 1e-4 relative to the scale."""
    import torch.nn.functional as F
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator(device="cpu").manual_seed(C_ + H + W)
    x = torch.randn(N, C_, H, W, generator=g).to(DEV)
    res = torch.randn(N, C_, H, W, generator=g).to(DEV)
    ka, kb = ((1, 3), (3, 1)) if axis == 1 else ((3, 1), (1, 3))
    wa = (torch.randn(C_, C_, *ka, generator=g) * (0.6 / C_ ** 0.5)).to(DEV)
    wb = (torch.randn(C_, C_, *kb, generator=g) * (0.6 / C_ ** 0.5)).to(DEV)
    ba, bb = (torch.randn(C_, generator=g) * 0.1).to(DEV), (torch.randn(C_, generator=g) * 0.1).to(DEV)

    def conv(t, w, b, d):
        kh = w.shape[2]
        return F.conv2d(t, w, b, padding=(d, 0) if kh == 3 else (0, d), dilation=(d, 1) if kh == 3 else (1, d))

    mid_ref = F.relu(conv(x, wa, ba, d1))
    y_ref = F.relu(conv(mid_ref, wb, bb, d2) + res)
    packed = hipf.conv_pair_pack(wa, wb)
    n0 = launches()
    y, mid = hipf.conv_pair(x, packed, ba, bb, axis, d1, d2, True, True, residual=res, keep_mid=True)
    y2 = hipf.conv_pair(x, packed, ba, bb, axis, d1, d2, True, True, residual=res)
    assert launches() - n0 == 2
    assert torch.equal(y, y2)
    gu.check_arrays(mid.cpu().numpy(), mid_ref.cpu().numpy(), 1e-4, 1e-4, what="mid", scale_rtol=1e-4)
    gu.check_arrays(y.cpu().numpy(), y_ref.cpu().numpy(), 1e-4, 1e-4, what="y", scale_rtol=1e-4)
    with pytest.raises(RuntimeError):
        hipf.conv_pair(x[:, :, :8, :8].contiguous(), packed, ba, bb, axis)


@pytest.mark.parametrize("C_,d,N,H,W", [(128, 4, 2, 64, 40), (128, 1, 1, 24, 130), (64, 1, 2, 128, 36),
                                        (64, 2, 1, 70, 200), (16, 1, 2, 40, 48)])
def test_alternative_kernel_paths_agree(C_, d, N, H, W):
    """The specialised kernels (line weight gradients for 64/128 channels, LDS-free kernels for 16
    channels) against the general implicit-GEMM / per-tap kernels on the same block: outputs equal
    bit for bit where only the launch plan differs, gradients to fp32 rounding of the sum
    (different summation order over ~10^4..10^5 pixels)."""
    from erfnet_pytorch_amd import functional as hipf, _lib
    g = torch.Generator(device="cpu").manual_seed(3 * C_ + H)
    x = torch.randn(N, C_, H, W, generator=g).clamp_(min=0).to(DEV).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(C_, C_, *k, generator=g) * (0.6 / C_ ** 0.5)).to(DEV).requires_grad_(True)
          for k in ((3, 1), (1, 3), (3, 1), (1, 3))]
    bs = [(torch.randn(C_, generator=g) * 0.1).to(DEV).requires_grad_(True) for _ in range(4)]
    gy = torch.randn(N, C_, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)

    def run():
        for q in ws + bs:
            q.grad = None
        xx = x.clone().requires_grad_(True)
        y = hipf.nb1d(xx, ws, bs, d)
        y.backward(gy)
        return [y.detach(), xx.grad] + [q.grad.clone() for q in ws + bs]

    opts = (_lib.OPT_WGRAD_LINE, _lib.OPT_STREAM16)
    prev = [_lib.set_option(o, 1) for o in opts]
    try:
        fast = run()
        for o in opts:
            _lib.set_option(o, 0)
        plain = run()
    finally:
        for o, v in zip(opts, prev):
            _lib.set_option(o, v)
    if C_ != 16:
        assert torch.equal(fast[0], plain[0]) and torch.equal(fast[1], plain[1])
    for i, (u, v) in enumerate(zip(fast, plain)):
        gu.check_arrays(u.cpu().numpy(), v.cpu().numpy(), 1e-5, 1e-4, what=f"tensor{i}", scale_rtol=2e-5)


def test_accuracy_metrics_match_reference_formulas():
    """ClassificationAccuracy / MeanAccuracy on the GPU (erf_accuracy_counts) against the reference's
    formulas (train/evaluation_functions.py:133-163) evaluated with torch ops: same accuracy, same
    per-class values (nan for absent classes), labels -1 ignored."""
    from erfnet_pytorch_amd.evaluation_functions import ClassificationAccuracy, MeanAccuracy
    g = torch.Generator().manual_seed(5)
    for n, k, h, w, present in ((2, 20, 64, 96, 20), (1, 5, 33, 47, 3), (3, 7, 16, 16, 7)):
        prob = torch.randn(n, k, h, w, generator=g)
        tgt = torch.randint(-1, present, (n, h, w), generator=g)
        # make some pixels certainly correct so the accuracy is not ~1/k everywhere
        idx = torch.rand(n, h, w, generator=g) < 0.4
        boost = torch.nn.functional.one_hot(tgt.clamp(min=0), k).permute(0, 3, 1, 2).float() * 8.0
        prob = prob + boost * idx.unsqueeze(1)
        n0 = launches()
        acc = ClassificationAccuracy()(prob.to(DEV), tgt.to(DEV))
        mean_acc, class_acc = MeanAccuracy(k)(prob.to(DEV), tgt.to(DEV))
        assert launches() - n0 == 4
        mx = prob.argmax(dim=1)
        ref_acc = (mx == tgt).sum().float() / (tgt >= 0).sum().float()
        ref_cls = np.array([float(((mx == tgt) & (tgt == c)).sum().float() / (tgt == c).sum().float()) for c in range(k)])
        assert abs(float(acc) - float(ref_acc)) < 1e-6
        np.testing.assert_allclose(class_acc, ref_cls, rtol=1e-6, equal_nan=True)
        assert abs(float(mean_acc) - float(np.nanmean(ref_cls))) < 1e-6


def test_loss_forward_leaves_accuracy_counts_for_the_metrics():
    """train/main.py:327-347 evaluates the loss and then both accuracy metrics on the same
    (logits, targets): the loss kernel counts on the side, the metrics launch nothing, and the
    values equal those of the stand-alone metric kernel; a changed tensor invalidates the counts."""
    from erfnet_pytorch_amd.evaluation_functions import ClassificationAccuracy, CrossEntropyLoss2d, MeanAccuracy
    g = torch.Generator().manual_seed(9)
    prob = torch.randn(2, 20, 48, 80, generator=g)
    tgt = torch.randint(-1, 18, (2, 48, 80), generator=g)
    prob = prob + 6.0 * torch.nn.functional.one_hot(tgt.clamp(min=0), 20).permute(0, 3, 1, 2) * \
        (torch.rand(2, 1, 48, 80, generator=g) < 0.5)
    prob, tgt = prob.to(DEV), tgt.to(DEV)
    ref_acc = ClassificationAccuracy()(prob, tgt)
    ref_mean, ref_cls = MeanAccuracy(20)(prob, tgt)
    loss = CrossEntropyLoss2d()(prob, tgt)
    n0 = launches()
    acc = ClassificationAccuracy()(prob, tgt)
    mean, cls = MeanAccuracy(20)(prob, tgt)
    assert launches() == n0
    assert float(acc) == float(ref_acc) and float(mean) == float(ref_mean)
    np.testing.assert_array_equal(cls, ref_cls)
    assert abs(float(loss) - float(torch.nn.functional.cross_entropy(prob, tgt, ignore_index=-1))) < 1e-5
    tgt[0, 0, 0] = 3                                   # in-place change: the counts are stale
    n0 = launches()
    ClassificationAccuracy()(prob, tgt)
    assert launches() - n0 == 2


def test_predict_labels_equals_argmax_of_logits():
    """ERFNet.predict_labels (erf_outconv_argmax) against max(1)[1] of the logits of the same
    network: identical labels, int64 [N,1,H,W]; the CPU path agrees as well."""
    net = gu.fill_state_dict(make_model("eval_erfnet_nobn_19"), tag="eval_erfnet_nobn_19").eval()
    x = torch.from_numpy(gu.det_uniform("argmax.x", (2, 3, 64, 96), 0.0, 1.0))
    with torch.no_grad():
        lab_cpu = net.predict_labels(x)
        net = net.to(DEV)
        logits = first(net(x.to(DEV)))
        lab = net.predict_labels(x.to(DEV))
    assert lab.dtype == torch.int64 and lab.shape == (2, 1, 64, 96)
    assert torch.equal(lab, logits.max(1)[1].unsqueeze(1))
    assert (lab.cpu() != lab_cpu).float().mean() < 1e-3      # CPU conv rounding may flip near-ties


def _golden_label_case(key, size):
    """(input, golden arg-max map [N,H,W], mask of the pixels whose golden top-2 margin is no tie)."""
    if size == "fullres":
        x = gu.det_uniform("net.x_fullres", (1, 3, 1024, 2048), 0.0, 1.0)
        return x, EXTRA["fullres/argmax_s2"], EXTRA["fullres/margin_s2"].astype(np.float32) > 2.5e-3, 2
    shape = (1, 3, 64, 128) if size == "small" else (1, 3, 512, 1024)
    x = gu.det_uniform("net.x_" + size, shape, 0.0, 1.0)
    gold = NETS[f"{key}/{size}/argmax"]
    if size == "big":
        return x, gold, NETS[f"{key}/big/margin"].astype(np.float32) > 2.5e-3, 1
    return x, gold, None, 1


LABEL_CASES = [("eval_erfnet_20", "small"), ("eval_erfnet_nobn_19", "small"), ("eval_erfnet_20", "big"),
               ("eval_erfnet_nobn_19", "fullres")]


@pytest.mark.parametrize("key,size", LABEL_CASES)
def test_predict_labels_against_golden_argmax(key, size):
    """ERFNet.predict_labels (the fused output-convolution + arg-max kernel) against the REFERENCE's
    arg-max map: identical class on every pixel that is not a tie (golden top-2 margin above the
    logit tolerance; for the small case, which stores no margin, the margin of the HIP logits)."""
    x, gold, ok, step = _golden_label_case(key, size)
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    with torch.no_grad():
        lab = net.predict_labels(t(x))
        if ok is None:
            top2 = first(net(t(x))).topk(2, dim=1).values
            ok = ((top2[:, 0] - top2[:, 1]) > 2e-3).cpu().numpy()
    lab = lab[:, 0, ::step, ::step].cpu().numpy()
    assert ok.mean() > 0.9
    assert np.array_equal(lab.astype(np.uint8)[ok], gold[ok])
    assert (lab.astype(np.uint8) != gold).mean() < 2e-3          # ties included: a handful at most


@pytest.mark.parametrize("key,size", LABEL_CASES[2:])
def test_iou_gate_predicted_labels_score_like_the_reference(key, size):
    """The evaluation gate end to end (eval/eval_iou.py:90-92 + eval/iouEval.py:21-75): label maps
    from HIP predict_labels and the reference's arg-max map, both scored by iouEval against the
    same ground truth (the golden map with 30 % of the pixels relabelled at random, class 19 =
    ignore).  Per-class IoU must agree within 0.1 (the gate); in practice it agrees to 1e-3."""
    from erfnet_pytorch_amd.iouEval import iouEval
    x, gold, _, step = _golden_label_case(key, size)
    ncls = 20
    noise = gu.det_labels(f"iougate.{key}.noise", gold.shape, 0, ncls).astype(np.int64)
    swap = gu.det_uniform(f"iougate.{key}.swap", gold.shape, 0.0, 1.0) < 0.3
    truth = t(np.where(swap, noise, gold.astype(np.int64))).unsqueeze(1)
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    with torch.no_grad():
        lab = net.predict_labels(t(x))[:, :, ::step, ::step].contiguous()
    scores = []
    for pred in (lab, t(gold.astype(np.int64)).unsqueeze(1)):
        ev = iouEval(ncls)                      # ignoreIndex 19, as the reference's default
        ev.addBatch(pred, truth)
        scores.append(ev.getIoU())
    (m_hip, c_hip), (m_ref, c_ref) = scores
    d = (c_hip - c_ref).abs().max().item()
    assert d <= 0.1 and abs(float(m_hip) - float(m_ref)) <= 0.1
    assert d <= 1e-3, d
    assert float(m_ref) > 0.1                   # a real score (random maps would give 1/39 per class)


def test_iou_eval_gpu_counts():
    """iouEval.addBatch on GPU label maps (erf_iou_add_batch): the reference's golden IoU, and exact
    agreement of tp/fp/fn with the one-hot formulation on random maps with and without an ignore
    class, accumulated over several batches."""
    from erfnet_pytorch_amd.iouEval import iouEval
    pred = t(gu.det_labels("iou.pred", (2, 1, 32, 48), 0, 20).astype(np.int64))
    lab = t(gu.det_labels("iou.lab", (2, 1, 32, 48), 0, 20).astype(np.int64))
    ev = iouEval(20)
    n0 = launches()
    ev.addBatch(pred, lab)
    assert launches() - n0 == 2
    mean, per_class = ev.getIoU()
    ref = gu.load_json("iou.json")
    np.testing.assert_allclose(per_class.numpy(), ref["per_class"], rtol=1e-12)
    assert abs(float(mean) - ref["mean"]) < 1e-12
    g = torch.Generator().manual_seed(11)
    for ncls, ign in ((20, 19), (7, 19), (2, 1)):
        a, b = iouEval(ncls, ign), iouEval(ncls, ign)
        for shape in ((3, 1, 65, 129), (1, 1, 512, 1024)):
            x = torch.randint(0, ncls, shape, generator=g)
            y = torch.randint(0, ncls, shape, generator=g)
            a.addBatch(x.to(DEV), y.to(DEV))
            b.addBatch(x, y)                       # CPU tensors: one-hot formulation
        for u, v in ((a.tp, b.tp), (a.fp, b.fp), (a.fn, b.fn)):
            assert torch.equal(u, v)
        a.reset()
        assert float(a.tp.sum()) == 0.0


# ------------------------------------------------------------- full benchmark size: properties
def _torch_nb1d(x, ws, bs, d):
    import torch.nn.functional as F
    o = F.relu(F.conv2d(x, ws[0], bs[0], padding=(1, 0)))
    o = F.relu(F.conv2d(o, ws[1], bs[1], padding=(0, 1)))
    o = F.relu(F.conv2d(o, ws[2], bs[2], padding=(d, 0), dilation=(d, 1)))
    o = F.conv2d(o, ws[3], bs[3], padding=(0, d), dilation=(1, d))
    return F.relu(o + x)


@pytest.mark.parametrize("C_,d,H,W", [(64, 1, 128, 256), (128, 8, 64, 128), (16, 1, 256, 512)])
def test_nb1d_full_size_vs_torch_fp32(C_, d, H, W):
    """Batch 6 at the benchmark's stage resolutions against stock torch fp32 ops on the same device
    (forward and all gradients; tolerance 1e-3 relative to each tensor's scale)."""
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(6, C_, H, W, generator=g).clamp_(min=0).to(DEV)
    shapes = [(C_, C_, 3, 1), (C_, C_, 1, 3), (C_, C_, 3, 1), (C_, C_, 1, 3)]
    bound = (1.0 / (3 * C_)) ** 0.5
    ws = [((torch.rand(s, generator=g) * 2 - 1) * bound * 1.7).to(DEV).requires_grad_(True) for s in shapes]
    bs = [((torch.rand(C_, generator=g) * 2 - 1) * 0.1).to(DEV).requires_grad_(True) for _ in range(4)]
    gy = torch.randn(6, C_, H, W, generator=g).to(DEV)
    x1 = x.clone().requires_grad_(True)
    y1 = hipf.nb1d(x1, ws, bs, d)
    y1.backward(gy)
    got = [y1.detach(), x1.grad] + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]
    for p in ws + bs:
        p.grad = None
    x2 = x.clone().requires_grad_(True)
    y2 = _torch_nb1d(x2, ws, bs, d)
    y2.backward(gy)
    ref = [y2.detach(), x2.grad] + [w.grad for w in ws] + [b.grad for b in bs]
    for i, (a, b) in enumerate(zip(got, ref)):
        a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
        # rare ReLU mask flips (pre-activation within rounding of 0) perturb a 3x3x3-pixel
        # neighbourhood of dx by O(|g| |w|); everything else must agree to 1e-3 of the scale
        # (parameter gradients sum over ~50k-800k pixels, so one flip moves an element by O(1):
        # they get a tolerance relative to the tensor's scale, and one element in 10^4 may miss it
        # by a few flips' worth)
        if i < 2:
            gu.check_arrays(a, b, 1e-4, 1e-3, what=f"tensor{i}", scale_rtol=1e-3, outlier_frac=0.002,
                            outlier_rtol=0.5)
        else:
            gu.check_arrays(a, b, 1e-4, 1e-3, what=f"tensor{i}", scale_rtol=5e-3, outlier_frac=1e-4,
                            outlier_rtol=2e-2)


def test_full_size_properties_batch6():
    """BASELINE configs[1] size (6x3x512x1024, 20 classes): run-to-run determinism, batch
    independence (image i of the batch equals the same image alone) and finite outputs."""
    net = gu.fill_state_dict(make_model("net_classification_20"), tag="net_classification_20").to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.rand(6, 3, 512, 1024, generator=g).to(DEV)
    with torch.no_grad():
        y_a = first(net(x))
        y_b = first(net(x))
        y_1 = first(net(x[2:3]))
    assert y_a.shape == (6, 20, 512, 1024)
    assert torch.isfinite(y_a).all()
    assert torch.equal(y_a, y_b)
    assert torch.allclose(y_a[2:3], y_1, atol=1e-5, rtol=1e-5)


def test_training_step_is_bitwise_reproducible():
    """Two identical forward/backward passes of the 20-class network at 2x3x256x512 give bit-identical
    logits and gradients: every reduction in the kernels runs in a fixed order (no atomics on
    floating-point data), and a race between blocks would show up here."""
    net = gu.fill_state_dict(make_model("net_classification_20"), tag="trainstep").to(DEV).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    g = torch.Generator().manual_seed(21)
    x = torch.rand(2, 3, 256, 512, generator=g).to(DEV)
    lab = torch.randint(-1, 20, (2, 256, 512), generator=g).to(DEV)
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        out = first(net(x))
        torch.nn.functional.cross_entropy(out, lab, ignore_index=-1).backward()
        runs.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 100
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


# --------------------------------------------------------------------------------- second fixture batch
EXTRA = gu.load_npz("extra.npz")
EXTRA_META = gu.load_json("extra.json")


@pytest.mark.timeout(300)
def test_fullres_1024x2048_erfnet_nobn():
    """The full-resolution forward of eval/eval_forwardTime.py's model (erfnet_nobn.ERFNet(19),
    1 x 3 x 1024 x 2048): logits within 1e-3, identical arg-max map away from ties."""
    key = "eval_erfnet_nobn_19"
    net = gu.fill_state_dict(make_model(key), tag=key).to(DEV).eval()
    x = t(gu.det_uniform("net.x_fullres", (1, 3, 1024, 2048), 0.0, 1.0))
    with torch.no_grad():
        y = net(x).cpu().numpy()
    gu.check(EXTRA, "fullres/y", y, 1e-3, 1e-3)
    np.testing.assert_allclose(y[:, :, 480:544, 992:1056], EXTRA["fullres/crop"], atol=1e-3, rtol=1e-3)
    ok = EXTRA["fullres/margin_s2"].astype(np.float32) > 2.5e-3
    assert ok.mean() > 0.9
    assert np.array_equal(y.argmax(1)[:, ::2, ::2].astype(np.uint8)[ok], EXTRA["fullres/argmax_s2"][ok])


@pytest.mark.timeout(300)
def test_imagenet_two_sgd_steps_bn_train():
    """imagenet/erfnet_imagenet.ERFNet(1000) with BatchNorm in train mode, two steps of the
    reference's SGD (imagenet/main.py:83-85).  The first step is compared tightly.  The second
    step runs on parameters that moved by lr 0.1 x (ill-conditioned) gradients: the reference's own
    fp32 and fp64 runs differ by 0.06 in that loss (extra.json), which sets the bound there."""
    from erfnet_pytorch_amd import erfnet_imagenet
    meta = EXTRA_META["imagenet_train"]
    net = gu.fill_state_dict(erfnet_imagenet.ERFNet(1000), tag="imagenet_train").to(DEV)
    for sub in net.modules():
        if isinstance(sub, torch.nn.Dropout2d):
            sub.p = 0
    net.train()
    opt = torch.optim.SGD(net.parameters(), 0.1, momentum=0.9, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss()
    x = t(gu.det_uniform("imagenet_train.x", (4, 3, 224, 224), 0.0, 1.0))
    target = t(gu.det_labels("imagenet_train.target", (4,), 0, 1000))
    losses, outs = [], []
    before = launches()
    for _ in range(2):
        out = net(x)
        loss = crit(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
        outs.append(out.detach().cpu().numpy())
    assert launches() > before
    assert abs(losses[0] - meta["losses"][0]) <= 1e-3
    gu.check_arrays(outs[0], EXTRA["imagenet_train/out0"], 1e-3, 1e-3, what="logits, step 1")
    gap = abs(meta["losses"][1] - meta["losses_fp64"][1])
    assert abs(losses[1] - meta["losses"][1]) <= max(2.0 * gap, 0.05), (losses, meta["losses"])
    ref1 = EXTRA["imagenet_train/out1"].astype(np.float64)
    ref1_64 = EXTRA["imagenet_train/out1_fp64"].astype(np.float64)
    drift = np.sqrt(((ref1 - ref1_64) ** 2).sum() / (ref1_64 ** 2).sum())
    err = np.sqrt(((outs[1] - ref1) ** 2).sum() / (ref1 ** 2).sum())
    assert err <= max(2.0 * drift, 2e-2), (err, drift)
    # parameters after two steps: abs-sum checksums.  The reference's own fp32 and fp64 runs differ
    # by 1-3e-2 (relative) on a dozen tensors and by 1.2e-1 on one (extra.json), so 3e-2 — or three
    # times that tensor's own fp32/fp64 gap — is what two correct implementations can be held to
    for k, p in net.named_parameters():
        a = meta["params"][k][1]
        drift_k = abs(a - meta["params_fp64"][k][1])
        got = float(p.detach().double().abs().sum())
        assert abs(got - a) <= max(3.0 * drift_k, 3e-2 * a) + 1e-4, (k, got, a, drift_k)
