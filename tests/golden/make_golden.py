#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the reference's own modules.

Run in the build container only (the reference lives at /root/reference and does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned (all with torch CPU fp32 kernels, the arithmetic the reference itself uses):
  blocks.npz     block-level forward outputs and gradients (DownsamplerBlock, non_bottleneck_1d,
                 UpsamplerBlock, output ConvTranspose2d, SoftMaxConv, LadderBlock, 1x1 output_conv)
                 for BN off / eval / train
  nets.npz       whole-network logits (eval/erfnet, eval/erfnet_nobn, train/net_* variants,
                 imagenet/erfnet_imagenet), argmax maps at 512x1024, BN-train forward/backward
  manifest.json  state_dict key -> shape per model, parameter counts, case list
  trainstep.json losses and parameter checksums after two Adam steps with the reference's settings
  iou.json       eval/iouEval.py on a seeded synthetic (prediction, label) pair
Inputs and weights are NOT stored: they are regenerated from tests/golden_util.py's keyed fills.
"""
import importlib.util
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_util as gu  # noqa: E402

REF = "/root/reference"
torch.manual_seed(0)
torch.set_num_threads(8)


def load_ref(subdir, fname, modname):
    """Import one reference source file under a private module name."""
    path = os.path.join(REF, subdir, fname)
    sys.path.insert(0, os.path.join(REF, subdir))
    try:
        for stale in ("erfnet_blocks",):
            sys.modules.pop(stale, None)
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.pop(0)
    return mod


class FixedDropout2d(torch.nn.Module):
    """Stands in for nn.Dropout2d with a known per-(n,c) keep mask (same arithmetic:
    zero whole feature maps, scale the rest by 1/(1-p))."""

    def __init__(self, p, scale):
        super().__init__()
        self.p = p
        self.scale = scale  # [N, C, 1, 1]

    def forward(self, x):
        return x * self.scale


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def set_bn_mode(mod, mode):
    """mode 0: BN off (flag false); 1: eval; 2: train."""
    if mode == 2:
        mod.train()
    else:
        mod.eval()


def run_case(store, name, mod, inputs, out_index=None):
    """Forward + backward of ``mod`` on keyed inputs; pack y, dx_i and every parameter grad."""
    xs = []
    for i, shp in enumerate(inputs):
        x = t(gu.det_normal(f"{name}.x{i}", shp))
        if name.endswith("relu_in"):
            x = torch.relu(x)
        x.requires_grad_(True)
        xs.append(x)
    y = mod(*xs)
    if out_index is not None:
        y = y[out_index]
    gy = t(gu.det_normal(f"{name}.gy", tuple(y.shape)))
    y.backward(gy)
    gu.pack(store, f"{name}/y", y.detach().numpy())
    for i, x in enumerate(xs):
        gu.pack(store, f"{name}/dx{i}", x.grad.numpy())
    for k, p in mod.named_parameters():
        if p.grad is not None:
            gu.pack(store, f"{name}/d:{k}", p.grad.numpy())
    for k, b in mod.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            gu.pack(store, f"{name}/buf:{k}", b.numpy())


def gen_blocks(manifest):
    blk = load_ref("train", "erfnet_blocks.py", "ref_train_blocks")
    store = {}
    cases = []

    def add(kind, name, ctor, inputs, bn_mode, **extra):
        mod = ctor()
        gu.fill_state_dict(mod, tag=name)
        set_bn_mode(mod, bn_mode)
        drop = extra.pop("drop_scale_key", None)
        if drop:
            N, C = inputs[0][0], inputs[0][1]
            keep = (gu.det_uniform(drop, (N, C, 1, 1), 0, 1) >= extra["p"]).astype(np.float32)
            mod.dropout = FixedDropout2d(extra["p"], t(keep / (1.0 - extra["p"])))
            mod.use_dropout = True
        run_case(store, name, mod, inputs)
        cases.append(dict(kind=kind, name=name, inputs=[list(s) for s in inputs], bn_mode=bn_mode,
                          drop_scale_key=drop, **extra))

    for bn in (0, 1, 2):
        ub = bn > 0
        for (ci, co) in ((3, 16), (16, 64), (64, 128)):
            add("down", f"down_{ci}_{co}_bn{bn}",
                lambda ci=ci, co=co: blk.DownsamplerBlock(ci, co, use_batchnorm=ub),
                [(2, ci, 16, 24)], bn, cin=ci, cout=co)
        for (ci, co) in ((128, 64), (64, 16)):
            add("up", f"up_{ci}_{co}_bn{bn}",
                lambda ci=ci, co=co: blk.UpsamplerBlock(ci, co, use_batchnorm=ub),
                [(2, ci, 6, 10)], bn, cin=ci, cout=co)
    # ties in the 2x2 max-pool windows (post-ReLU input, as inside the real network)
    add("down", "down_16_64_bn0_relu_in", lambda: blk.DownsamplerBlock(16, 64),
        [(2, 16, 16, 24)], 0, cin=16, cout=64)

    nb_cases = [(16, 1, 2, (0, 1, 2)), (16, 2, 2, (0,)), (64, 1, 2, (0,)), (64, 4, 2, (0, 1, 2)),
                (64, 8, 2, (0,)), (128, 2, 1, (0,)), (128, 16, 1, (0, 1, 2)), (32, 1, 2, (0,))]
    for (c, d, n, modes) in nb_cases:
        for bn in modes:
            add("nb1d", f"nb1d_{c}_d{d}_bn{bn}",
                lambda c=c, d=d, bn=bn: blk.non_bottleneck_1d(c, 0.0, d, use_batchnorm=bn > 0),
                [(n, c, 20, 24)], bn, chann=c, dilated=d)
    # Dropout2d with an injected keep mask (train mode, BN on and off)
    add("nb1d", "nb1d_64_d2_bn0_drop",
        lambda: blk.non_bottleneck_1d(64, 0.3, 2, use_dropout=True), [(2, 64, 20, 24)], 0,
        chann=64, dilated=2, p=0.3, drop_scale_key="nb1d_64_d2_bn0_drop.keep")
    add("nb1d", "nb1d_64_d2_bn2_drop",
        lambda: blk.non_bottleneck_1d(64, 0.3, 2, use_dropout=True, use_batchnorm=True),
        [(2, 64, 20, 24)], 2, chann=64, dilated=2, p=0.3,
        drop_scale_key="nb1d_64_d2_bn2_drop.keep")

    # output ConvTranspose2d(16, C, 2, stride=2) (train/net_classification.py:56) and SoftMaxConv
    for ncls in (20, 1, 2):
        add("outconv", f"outconv_16_{ncls}",
            lambda ncls=ncls: torch.nn.ConvTranspose2d(16, ncls, 2, stride=2, padding=0,
                                                       output_padding=0, bias=True),
            [(2, 16, 6, 10)], 0, cin=16, ncls=ncls)
    add("softmaxconv", "softmaxconv_16_20", lambda: blk.SoftMaxConv(16, 20, 0.1, False),
        [(2, 16, 6, 10)], 0, cin=16, ncls=20)
    # Encoder.output_conv: 1x1 conv 128 -> classes (eval/erfnet.py:85)
    add("conv1x1", "conv1x1_128_20", lambda: torch.nn.Conv2d(128, 20, 1, stride=1, padding=0,
                                                            bias=True),
        [(2, 128, 6, 10)], 0, cin=128, cout=20)
    # LadderBlock (train/erfnet_blocks.py:140-154)
    for bn in (0,):
        for c in (16, 64):
            add("ladder", f"ladder_{c}_bn{bn}", lambda c=c: blk.LadderBlock(c),
                [(2, c, 12, 20), (2, c, 12, 20)], bn, noutput=c)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **store)
    manifest["block_cases"] = cases


def sd_manifest(mod):
    """Ordered [key, shape] pairs: the order is part of the surface (optimizer state is indexed)."""
    return [[k, list(v.shape)] for k, v in mod.state_dict().items()]


def n_params(mod):
    return int(sum(p.numel() for p in mod.parameters()))


def gen_nets(manifest):
    store = {}
    nets = {}
    ev = load_ref("eval", "erfnet.py", "ref_eval_erfnet")
    nobn = load_ref("eval", "erfnet_nobn.py", "ref_eval_erfnet_nobn")
    ncl = load_ref("train", "net_classification.py", "ref_net_classification")
    nrg = load_ref("train", "net_regression.py", "ref_net_regression")
    nrs = load_ref("train", "net_regression_small.py", "ref_net_regression_small")
    nrl = load_ref("train", "net_regression_ladder.py", "ref_net_regression_ladder")
    nrls = load_ref("train", "net_regression_ladder_small.py", "ref_net_regression_ladder_small")
    ncll = load_ref("train", "net_classification_ladder.py", "ref_net_classification_ladder")
    img = load_ref("imagenet", "erfnet_imagenet.py", "ref_erfnet_imagenet")

    x_small = t(gu.det_uniform("net.x_small", (1, 3, 64, 128), 0.0, 1.0))
    x_big = t(gu.det_uniform("net.x_big", (1, 3, 512, 1024), 0.0, 1.0))

    def first(o):
        return o[0] if isinstance(o, (tuple, list)) else o

    def record(name, mod, extra_small=None):
        gu.fill_state_dict(mod, tag=name)
        mod.eval()
        nets[name] = dict(state_dict=sd_manifest(mod), n_params=n_params(mod))
        with torch.no_grad():
            y = first(mod(x_small))
            gu.pack(store, f"{name}/small/y", y.numpy())
            store[f"{name}/small/argmax"] = y.argmax(1).numpy().astype(np.uint8)
            if extra_small:
                extra_small(mod)

    def record_big(name, mod):
        with torch.no_grad():
            y = first(mod(x_big))
        gu.pack(store, f"{name}/big/y", y.numpy())
        store[f"{name}/big/crop"] = y[:, :, 224:288, 480:544].numpy().copy()
        # top-2 margin lets the test exclude numerically tied pixels from the argmax comparison
        top2 = y.topk(2, dim=1).values
        store[f"{name}/big/argmax"] = y.argmax(1).numpy().astype(np.uint8)
        store[f"{name}/big/margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)

    m = ev.ERFNet(20)

    def enc_only(mod):
        ye = mod(x_small, only_encode=True)
        gu.pack(store, "eval_erfnet_20/small/y_encode", ye.numpy())
    record("eval_erfnet_20", m, enc_only)
    record_big("eval_erfnet_20", m)

    m = nobn.ERFNet(19)
    record("eval_erfnet_nobn_19", m)

    m = ncl.Net(softmax_classes=20)
    record("net_classification_20", m)
    record_big("net_classification_20", m)

    record("net_regression_0", nrg.Net())
    record("net_regression_0_lik", nrg.Net(likelihood_loss=True))
    record("net_regression_20", nrg.Net(softmax_classes=20))
    record("net_regression_small_0", nrs.Net())
    record("net_regression_ladder_0", nrl.Net())
    record("net_regression_ladder_small_0", nrls.Net())
    record("net_classification_ladder_20", ncll.Net(softmax_classes=20))

    m = img.ERFNet(1000)
    gu.fill_state_dict(m, tag="imagenet_1000")
    m.eval()
    nets["imagenet_1000"] = dict(state_dict=sd_manifest(m), n_params=n_params(m))
    xi = t(gu.det_uniform("net.x_imagenet", (2, 3, 224, 224), 0.0, 1.0))
    with torch.no_grad():
        gu.pack(store, "imagenet_1000/y", m(xi).numpy())

    # BN in training mode through the whole network (dropout probability forced to 0 because
    # the RNG stream of nn.Dropout2d cannot be reproduced by another implementation).
    m = ev.ERFNet(20)
    gu.fill_state_dict(m, tag="eval_erfnet_20_train")
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout2d):
            sub.p = 0
    m.train()
    xt = t(gu.det_uniform("net.x_train", (2, 3, 64, 128), 0.0, 1.0))
    lab = t(gu.det_labels("net.lab_train", (2, 64, 128), -1, 20))
    out = m(xt)
    loss = torch.nn.CrossEntropyLoss(ignore_index=-1)(out, lab)
    loss.backward()
    gu.pack(store, "eval_erfnet_20_train/y", out.detach().numpy())
    store["eval_erfnet_20_train/loss"] = np.array([loss.item()], np.float64)
    for k, p in m.named_parameters():
        if p.grad is not None:
            gu.pack(store, f"eval_erfnet_20_train/d:{k}", p.grad.numpy())
    for k, b in m.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            gu.pack(store, f"eval_erfnet_20_train/buf:{k}", b.numpy())

    np.savez_compressed(os.path.join(HERE, "nets.npz"), **store)
    manifest["nets"] = nets
    return ncl


def gen_trainstep(ncl):
    """Two optimiser steps exactly as train/main.py:43-51,213 sets them up
    (Adam lr 1e-4, betas (0.9, 0.999), eps 1e-4, weight_decay 1e-6; CrossEntropy ignore -1)."""
    m = ncl.Net(softmax_classes=20)
    gu.fill_state_dict(m, tag="trainstep")
    m.train()
    opt = torch.optim.Adam(m.parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)
    crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
    x = t(gu.det_uniform("trainstep.x", (2, 3, 64, 128), 0.0, 1.0))
    lab = t(gu.det_labels("trainstep.lab", (2, 64, 128), -1, 20))
    rec = dict(losses=[], params={})
    for _ in range(2):
        opt.zero_grad()
        out, _cp = m(x)
        loss = crit(out, lab)
        loss.backward()
        opt.step()
        rec["losses"].append(float(loss.item()))
    for k, p in m.named_parameters():
        a = p.detach().double()
        rec["params"][k] = [float(a.sum()), float(a.abs().sum())]
    with open(os.path.join(HERE, "trainstep.json"), "w") as f:
        json.dump(rec, f, indent=0)


def gen_iou():
    io = load_ref("eval", "iouEval.py", "ref_ioueval")
    pred = t(gu.det_labels("iou.pred", (2, 1, 32, 48), 0, 20))
    lab = t(gu.det_labels("iou.lab", (2, 1, 32, 48), 0, 20))
    ev = io.iouEval(20)
    ev.addBatch(pred, lab)
    mean, per_class = ev.getIoU()
    with open(os.path.join(HERE, "iou.json"), "w") as f:
        json.dump(dict(mean=float(mean), per_class=[float(v) for v in per_class]), f)


def main():
    manifest = {"torch": torch.__version__, "numpy": np.__version__}
    gen_blocks(manifest)
    ncl = gen_nets(manifest)
    gen_trainstep(ncl)
    gen_iou()
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
