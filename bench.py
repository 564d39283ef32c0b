#!/usr/bin/env python3
"""Benchmark of the ERFNet training step on MI355X (BASELINE.json metric: images/sec train,
512x1024, 20 classes).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + CrossEntropy(ignore -1) + backward + Adam update (the reference's settings,
train/main.py:43-51,213) of net_classification.Net(softmax_classes=20) on a synthetic batch of
6 x 3 x 512 x 1024 per GPU, fp32, inputs resident in HBM.  With N > 1 every rank holds a replica and
the gradients are averaged over RCCL (erfnet_pytorch_amd/ddp.py); per-GPU work is fixed (weak
scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PER_GPU_BATCH, HEIGHT, WIDTH, CLASSES = 6, 512, 1024, 20
# dominant kernel: the implicit-GEMM 1-D convolution at the 128-channel stage
DOM_C, DOM_H, DOM_W = 128, HEIGHT // 8, WIDTH // 8
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU")
    ap.add_argument("--model", default="net_classification",
                    choices=["net_classification", "erfnet_bn"],
                    help="net_classification: the fork's training model (BatchNorm off); "
                         "erfnet_bn: eval/erfnet.ERFNet(20) with BatchNorm + Dropout2d in train mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="also report forward ms/img only")
    return ap.parse_args()


def build_model(name):
    if name == "net_classification":
        from erfnet_pytorch_amd import net_classification
        return net_classification.Net(softmax_classes=CLASSES)
    from erfnet_pytorch_amd import erfnet
    return erfnet.ERFNet(CLASSES)


def logits_of(out):
    return out[0] if isinstance(out, (tuple, list)) else out


def dominant_kernel_roofline(device, batch):
    """Time the dominant kernel with HIP events on the stream it is launched on.

    pair_f32_kernel<64,128> — two chained 3-tap convolutions 128 -> 128 channels over
    batch x 64 x 128 pixels in one launch (conv1x3 then conv3x1, intermediate stored as in a
    training forward) — takes the largest share of a training step.  erf_conv_pair_forward issues
    exactly one such launch per call (weights are packed once, outside the timed region);
    algorithmic FLOPs per launch = 2 convolutions x 2 * pixels * (3*128) * 128, algorithmic bytes =
    input read + intermediate written + output written."""
    from erfnet_pytorch_amd import functional as hipf
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(batch, DOM_C, DOM_H, DOM_W, generator=g).clamp_(min=0).to(device)
    x = x.contiguous(memory_format=torch.channels_last)
    wa = (torch.randn(DOM_C, DOM_C, 1, 3, generator=g) * 0.05).to(device)
    wb = (torch.randn(DOM_C, DOM_C, 3, 1, generator=g) * 0.05).to(device)
    b = torch.zeros(DOM_C, device=device)
    packed = hipf.conv_pair_pack(wa, wb)
    stream = torch.cuda.current_stream(device)
    reps = 50
    for _ in range(5):
        hipf.conv_pair(x, packed, b, b, 1, 2, 2, keep_mid=True)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(reps):
        hipf.conv_pair(x, packed, b, b, 1, 2, 2, keep_mid=True)
    stop.record(stream)
    stop.synchronize()
    ms_per_launch = start.elapsed_time(stop) / reps
    pixels = batch * DOM_H * DOM_W
    flops = 2 * 2.0 * pixels * (3 * DOM_C) * DOM_C
    nbytes = 3 * pixels * DOM_C * 4
    achieved = flops / (ms_per_launch * 1e-3) / 1e12
    traffic = None
    tj = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    if os.path.exists(tj):
        try:
            traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic,
            "kernel": "pair_f32_kernel<64,128> (conv1x3 + conv3x1, 128->128 channels, one launch)",
            "us_per_launch": round(ms_per_launch * 1e3, 2), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": nbytes,
            "hbm_bound_us": round(nbytes / 8e12 * 1e6, 2)}


def forward_time_harness(device, batch=1, iters=20):
    """The reference's forward-time script (eval/eval_forwardTime.py:22-58): erfnet_nobn.ERFNet(19)
    in eval mode on a fixed torch.randn(B,3,512,1024), device synchronised around every call, the
    first call excluded, seconds per image = t / B (the script's sleep(1) between calls dropped)."""
    from erfnet_pytorch_amd import erfnet_nobn
    model = erfnet_nobn.ERFNet(19).to(device).eval()
    images = torch.randn(batch, 3, HEIGHT, WIDTH, generator=torch.Generator().manual_seed(0)).to(device)
    times = []
    with torch.no_grad():
        for i in range(iters + 1):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            model(images)
            torch.cuda.synchronize(device)
            if i:
                times.append(time.perf_counter() - t0)
    out = {"model": "erfnet_nobn.ERFNet(19)", "batch": batch, "input": "randn 3x512x1024, fp32",
           "ms_per_img_mean": round(sum(times) / len(times) / batch * 1e3, 4),
           "ms_per_img_min": round(min(times) / batch * 1e3, 4)}
    # the same loop with the forward captured once into a HIP graph (one launch per call)
    from erfnet_pytorch_amd.graphs import GraphedForward
    graphed = GraphedForward(model, images)
    times = []
    for i in range(iters + 1):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        graphed(images)
        torch.cuda.synchronize(device)
        if i:
            times.append(time.perf_counter() - t0)
    out["hip_graph_ms_per_img_mean"] = round(sum(times) / len(times) / batch * 1e3, 4)
    out["hip_graph_ms_per_img_min"] = round(min(times) / batch * 1e3, 4)
    return out


def cpu_baseline():
    """The plain-C oracle (a port of the reference's arithmetic) timed on this host's cores:
    forward + CrossEntropy + backward of ONE 512x1024 image (no optimiser step)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle import oracle as orc
    model = build_model("net_classification")
    state = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(0)
    x = rng.random((1, 3, HEIGHT, WIDTH), dtype=np.float32)
    lab = rng.integers(-1, CLASSES, (1, HEIGHT, WIDTH))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    t0 = time.perf_counter()
    orc.erfnet_forward(x, state, "train")
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.erfnet_train_pass(x, lab, state, "train")
    t_train = time.perf_counter() - t0
    out = {"value": round(1.0 / t_train, 4), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "1 image 512x1024: forward + CrossEntropy + backward in the C oracle (OpenMP), "
                     "no optimiser step; forward alone %.0f ms/img" % (t_fwd * 1e3)}
    # the same arithmetic through stock torch CPU ops (what the reference itself runs on a CPU)
    model.train()
    torch.set_num_threads(cores)
    xt, lt = torch.from_numpy(x), torch.from_numpy(lab)
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    crit = CrossEntropyLoss2d()       # train/main.py:186-188
    with torch.no_grad():
        model(xt)
        t0 = time.perf_counter()
        model(xt)
        tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    crit(logits_of(model(xt)), lt).backward()
    tt = time.perf_counter() - t0
    out["torch_cpu_path"] = {"fwd_ms_per_img": round(tf * 1e3, 1), "train_images_per_s": round(1.0 / tt, 3),
                             "threads": cores}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("ERFNET_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE is {world}; using {world}", file=sys.stderr)
    local_rank %= max(torch.cuda.device_count(), 1)     # rehearsals with more ranks than GPUs share devices
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from erfnet_pytorch_amd import _lib
    from erfnet_pytorch_amd.ddp import GradientReducer
    _lib.lib()                      # fail loudly here if the HIP library is missing

    torch.manual_seed(0)
    model = build_model(args.model).to(device).train()
    # three buckets for the 20-class network (3.9 + 3.7 + 0.6 MB): the last one, the only one whose
    # all-reduce cannot hide behind backward, is the small remainder
    reducer = GradientReducer(model, bucket_bytes=3_600_000, direct_write=True)
    reducer.broadcast_parameters(0)
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    crit = CrossEntropyLoss2d()       # train/main.py:186-188

    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    images = torch.rand(args.batch, 3, HEIGHT, WIDTH, generator=g).to(device)
    labels = torch.randint(-1, CLASSES, (args.batch, HEIGHT, WIDTH), generator=g).to(device)

    # Set-up pass (no update): one backward tells the reducer which parameters take part; they are
    # then laid out in flat buffers next to their gradient buckets.  The optimiser keeps the
    # reference's settings (train/main.py:43-46,213); on the flat buffers torch's fused Adam is the
    # same element-wise update in one launch per bucket, and parameters that never get a gradient
    # are skipped as Adam skips them.
    reducer.zero_grad()
    crit(logits_of(model(images)), labels).backward()
    reducer.finish()
    params = reducer.flatten_parameters()
    opt = torch.optim.Adam(params, 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6, fused=True)

    def step():
        reducer.zero_grad()
        loss = crit(logits_of(model(images)), labels)
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    launches0 = _lib.launch_count()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert _lib.launch_count() > launches0, "the HIP kernels did not run"
    final_loss = float(loss.item())

    fwd_ms = None
    if rank == 0:
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(images)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(5):
                model(images)
            torch.cuda.synchronize(device)
            fwd_ms = (time.perf_counter() - t1) / 5 / args.batch * 1e3

    h1 = forward_time_harness(device) if rank == 0 else None

    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        rec = {
            "metric": "images/sec train (512x1024, 20cls)",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (uniform [0,1) images, labels in [-1,20), random-init weights)",
            "config": {"workload": "1xMI355X training, batch=6, 512x1024, 20 classes, fp32, "
                                   "encoder+decoder (BASELINE configs[1])",
                       "model": args.model, "per_gpu_batch": args.batch,
                       "global_batch": world * args.batch,
                       "step": "forward + CrossEntropy(ignore -1) + backward + Adam(lr 1e-4, eps 1e-4, wd 1e-6)",
                       "parallelism": f"dp{world}",
                       "grad_buckets_bytes": reducer.bucket_sizes},
            "fwd_ms_per_img": round(fwd_ms, 4),
            "forward_time_harness": h1,
            "final_loss": round(final_loss, 5),
        }
        if world == 1:
            rec["roofline"] = dominant_kernel_roofline(device, args.batch)
            if not args.no_cpu_baseline:
                rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
