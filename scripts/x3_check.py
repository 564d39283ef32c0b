"""Accuracy and speed of the split-bf16 path (ERF_OPT_SPLIT_BF16) against the fp32-MFMA path, both
measured against a float64 CPU evaluation (autograd) of the same block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from erfnet_pytorch_amd import _lib, erfnet_blocks

CONVS = ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")

def ref64(blk, x, gy):
    x = x.double().cpu().requires_grad_(True)
    P = {n: p.detach().double().cpu().requires_grad_(True) for n, p in blk.named_parameters()}
    def c(name, v):
        m = getattr(blk, name)
        return F.conv2d(v, P[name + ".weight"], P[name + ".bias"], padding=m.padding, dilation=m.dilation)
    def bn(name, v):
        m = getattr(blk, name)
        return F.batch_norm(v, m.running_mean.double().cpu(), m.running_var.double().cpu(), P[name + ".weight"], P[name + ".bias"], False, 0.0, m.eps)
    o = F.relu(c("conv3x1_1", x)); o = F.relu(bn("bn1", c("conv1x3_1", o)))
    o = F.relu(c("conv3x1_2", o)); o = bn("bn2", c("conv1x3_2", o))
    y = F.relu(o + x)
    y.backward(gy.double().cpu())
    return y.detach(), x.grad, {n: p.grad for n, p in P.items()}

def run(blk, x, gy, opt):
    _lib.lib().erf_set_option(_lib.OPT_SPLIT_BF16, 2 * opt)
    blk.zero_grad(set_to_none=True)
    xt = x.clone().requires_grad_(True)
    y = blk(xt); y.backward(gy)
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        xt.grad = None; blk(xt).backward(gy)
    t0.record()
    for _ in range(10):
        xt.grad = None; blk(xt).backward(gy)
    t1.record()
    torch.cuda.synchronize()
    blk.zero_grad(set_to_none=True)
    return y.detach(), xt.grad.clone(), g, t0.elapsed_time(t1) / 10

for (c, d, shape) in ((128, 2, (6, 128, 64, 128)), (64, 1, (6, 64, 128, 256)), (128, 16, (2, 128, 28, 50)), (64, 1, (1, 64, 37, 150))):
    torch.manual_seed(1)
    blk = erfnet_blocks.non_bottleneck_1d(c, 0.0, d, use_batchnorm=True).cuda().eval()
    with torch.no_grad():
        for m in (blk.bn1, blk.bn2):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
    x = torch.randn(shape, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
    y64, dx64, g64 = ref64(blk, x, gy)
    res = [run(blk, x, gy, 0), run(blk, x, gy, 1)]
    _lib.lib().erf_set_option(_lib.OPT_SPLIT_BF16, 1)
    print(f"C={c} d={d} {shape}: fwd+bwd ms fp32 {res[0][3]:.4f} split {res[1][3]:.4f}")
    for nm, (y, dx, g, _) in zip(("fp32 ", "split"), res):
        ey = (y.double().cpu() - y64).abs()
        e = (dx.double().cpu() - dx64).abs()
        big = e > 1e-4
        line = f"   [{nm}] y max err {ey.max().item():.3e} rms {ey.pow(2).mean().sqrt().item():.3e} | dx: {int(big.sum())} mask flips, rms of the rest {e[~big].pow(2).mean().sqrt().item():.3e}"
        rel = []
        for n in CONVS:
            for k in ("weight", "bias"):
                a, b = g[f"{n}.{k}"].double().cpu(), g64[f"{n}.{k}"]
                rel.append(((a - b).abs().max() / b.abs().max()).item())
        print(line + " | dw,db max err / max: " + " ".join(f"{r:.1e}" for r in rel))
