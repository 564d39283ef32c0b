#!/usr/bin/env python3
"""Second batch of golden fixtures (tests/golden/extra.npz, extra.json), generated like the first
by importing the reference's own modules in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py

  fullres/*      eval/erfnet_nobn.ERFNet(19) on a 1 x 3 x 1024 x 2048 input (the shape of
                 eval/eval_forwardTime.py's full-resolution run): logits (samples + statistics), a
                 64 x 64 crop, and arg-max class map + top-2 margin on every second pixel
  imagenet_train imagenet/erfnet_imagenet.ERFNet(1000), BatchNorm in train mode, two steps of
                 SGD(lr 0.1, momentum 0.9, weight decay 1e-4) with CrossEntropyLoss as
                 imagenet/main.py:83-85,168-189 sets them up (Dropout2d probability forced to 0: its
                 RNG stream cannot be reproduced); losses, logits of both steps and per-parameter
                 checksums, in fp32 and — to show how far two correct fp32 implementations may
                 drift on this ill-conditioned problem — in fp64
  losses/*       the regression / consistency losses of train/evaluation_functions.py:16-131 on
                 seeded inputs: values and gradients
make_golden.py's files are not touched.
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_util as gu  # noqa: E402
from make_golden import load_ref, t  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def gen_fullres(store):
    nobn = load_ref("eval", "erfnet_nobn.py", "ref_eval_erfnet_nobn")
    m = nobn.ERFNet(19)
    gu.fill_state_dict(m, tag="eval_erfnet_nobn_19")
    m.eval()
    x = t(gu.det_uniform("net.x_fullres", (1, 3, 1024, 2048), 0.0, 1.0))
    with torch.no_grad():
        y = m(x)
    gu.pack(store, "fullres/y", y.numpy())
    store["fullres/crop"] = y[:, :, 480:544, 992:1056].numpy().copy()
    top2 = y.topk(2, dim=1).values
    store["fullres/argmax_s2"] = y.argmax(1)[:, ::2, ::2].numpy().astype(np.uint8)
    store["fullres/margin_s2"] = (top2[:, 0] - top2[:, 1])[:, ::2, ::2].numpy().astype(np.float16)


def imagenet_steps(img, dtype):
    m = img.ERFNet(1000)
    gu.fill_state_dict(m, tag="imagenet_train")
    m = m.to(dtype)
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout2d):
            sub.p = 0
    m.train()
    opt = torch.optim.SGD(m.parameters(), 0.1, momentum=0.9, weight_decay=1e-4)     # imagenet/main.py:83-85
    crit = torch.nn.CrossEntropyLoss()
    x = t(gu.det_uniform("imagenet_train.x", (4, 3, 224, 224), 0.0, 1.0)).to(dtype)
    target = t(gu.det_labels("imagenet_train.target", (4,), 0, 1000))
    losses, outs = [], []
    for _ in range(2):
        out = m(x)
        loss = crit(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
        outs.append(out.detach().double().numpy())
    params = {k: [float(p.detach().double().sum()), float(p.detach().double().abs().sum())]
              for k, p in m.named_parameters()}
    bufs = {k: [float(b.double().sum()), float(b.double().abs().sum())] for k, b in m.named_buffers()
            if k.endswith("running_mean") or k.endswith("running_var")}
    return losses, outs, params, bufs


def gen_imagenet(store, meta):
    img = load_ref("imagenet", "erfnet_imagenet.py", "ref_erfnet_imagenet")
    l32, o32, p32, b32 = imagenet_steps(img, torch.float32)
    l64, o64, p64, b64 = imagenet_steps(img, torch.float64)
    for i in range(2):
        store[f"imagenet_train/out{i}"] = o32[i].astype(np.float32)
        store[f"imagenet_train/out{i}_fp64"] = o64[i].astype(np.float32)
    meta["imagenet_train"] = dict(losses=l32, losses_fp64=l64, params=p32, params_fp64=p64, buffers=b32,
                                  buffers_fp64=b64)


def gen_losses(store, meta):
    ef = load_ref("train", "evaluation_functions.py", "ref_evaluation_functions")
    n, k, h, w = 2, 5, 24, 40
    rec = {}

    def leaf(a):
        return t(a).clone().requires_grad_(True)

    targets = gu.det_uniform("loss.targets", (n, h, w), 0.0, 3.0)
    targets[gu.det_uniform("loss.mask", (n, h, w), 0.0, 1.0) < 0.3] = -1.0       # unlabelled pixels
    tg = t(targets)
    store["losses/targets"] = targets

    # L1LossMasked (evaluation_functions.py:89-97)
    o = leaf(gu.det_uniform("loss.l1.out", (n, 1, h, w), 0.0, 3.0))
    loss = ef.L1LossMasked()(o, tg)
    loss.backward()
    rec["l1_masked"] = float(loss.item())
    store["losses/l1_masked/d_out"] = o.grad.numpy()

    # LogLikelihoodLossMasked (:71-87), opt_eps as train/main.py:47,183 passes it
    c = leaf(gu.det_uniform("loss.ll.cost", (n, h, w), 0.0, 3.0))
    v = leaf(gu.det_uniform("loss.ll.var", (n, h, w), 0.05, 2.0))
    loss = ef.LogLikelihoodLossMasked(opt_eps=1e-4)(c, v, tg)
    loss.backward()
    rec["ll_masked"] = float(loss.item())
    store["losses/ll_masked/d_cost"] = c.grad.numpy()
    store["losses/ll_masked/d_var"] = v.grad.numpy()

    # L1LossClassProbMasked (:32-47): class probabilities x per-class cost [1,K,1,1]
    prob = torch.softmax(t(gu.det_normal("loss.cp.logits", (n, k, h, w))), dim=1).detach().clone().requires_grad_(True)
    cost = leaf(gu.det_uniform("loss.cp.cost", (1, k, 1, 1), 0.0, 3.0))
    loss = ef.L1LossClassProbMasked()(prob, cost, tg)
    loss.backward()
    rec["l1_classprob"] = float(loss.item())
    store["losses/l1_classprob/d_prob"] = prob.grad.numpy()
    store["losses/l1_classprob/d_cost"] = cost.grad.numpy()

    # LogLikelihoodLossClassProbMasked (:49-69)
    prob2 = prob.detach().clone().requires_grad_(True)
    cost2 = leaf(gu.det_uniform("loss.cp.cost", (1, k, 1, 1), 0.0, 3.0))
    var2 = leaf(gu.det_uniform("loss.cp.var", (1, k, 1, 1), 0.05, 2.0))
    loss = ef.LogLikelihoodLossClassProbMasked()(prob2, cost2, var2, tg)
    loss.backward()
    rec["ll_classprob"] = float(loss.item())
    store["losses/ll_classprob/d_prob"] = prob2.grad.numpy()
    store["losses/ll_classprob/d_cost"] = cost2.grad.numpy()
    store["losses/ll_classprob/d_var"] = var2.grad.numpy()

    # MSELossWeighted (:121-131), the mean-teacher consistency loss (train/main.py:190,352,359)
    a = leaf(gu.det_uniform("loss.mse.a", (n, k, h, w), 0.0, 1.0))
    b = t(gu.det_uniform("loss.mse.b", (n, k, h, w), 0.0, 1.0))
    loss = ef.MSELossWeighted(2.5)(a, b, 0.37)
    loss.backward()
    rec["mse_weighted"] = float(loss.item())
    store["losses/mse_weighted/d_out"] = a.grad.numpy()
    meta["losses"] = dict(values=rec, shape=[n, k, h, w], scale_factor=2.5, weight=0.37, opt_eps=1e-4)


def main():
    store, meta = {}, {"torch": torch.__version__}
    gen_fullres(store)
    gen_imagenet(store, meta)
    gen_losses(store, meta)
    np.savez_compressed(os.path.join(HERE, "extra.npz"), **store)
    with open(os.path.join(HERE, "extra.json"), "w") as f:
        json.dump(meta, f, indent=0)
    for fn in ("extra.npz", "extra.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))
    it = meta["imagenet_train"]
    print("imagenet losses fp32", it["losses"], "fp64", it["losses_fp64"])
    print("loss values", meta["losses"]["values"])


if __name__ == "__main__":
    main()
