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

--dtype bf16 runs the same step with bf16 activations (fp32 master parameters, fp32 gradients and
all-reduce).  At one GPU the line also carries, as extra keys, the bf16 training rate, the
full-resolution bf16 inference time of erfnet_nobn (the reference's eval_forwardTime.py loop at
1 x 3 x 1024 x 2048) with the bandwidth its non_bottleneck_1d blocks reach, and the ImageNet
pre-training step of imagenet/erfnet_imagenet.py.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PER_GPU_BATCH, HEIGHT, WIDTH, CLASSES = 6, 512, 1024, 20
# shape of the 128-channel stage (the bf16 dominant kernel's)
DOM_C, DOM_H, DOM_W = 128, HEIGHT // 8, WIDTH // 8
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
BF16_MATRIX_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0
# Algorithmic work per image: forward 53.2 GFLOP and 341.8 MB of fp32 block inputs + outputs; a
# training step is three times that (backward reads a block's input, its output or mask and the
# incoming gradient and writes the outgoing one).
FWD_GFLOP_PER_IMAGE, FWD_MB_PER_IMAGE_F32 = 53.2, 341.8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of about half a second (50 steps of 8-9 ms) so that run-to-run noise
    # stays well under one per cent
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU")
    ap.add_argument("--model", default="net_classification",
                    choices=["net_classification", "erfnet_bn"],
                    help="net_classification: the fork's training model (BatchNorm off); "
                         "erfnet_bn: eval/erfnet.ERFNet(20) with BatchNorm + Dropout2d in train mode")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage type of the activations (parameters and their gradients stay fp32)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra keys (bf16 rates, harnesses)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the matrix-pipe calibration (a child program; use under a profiler, whose "
                         "preloaded library has initialised the GPU before this script starts)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (for --gpus N > 1 without a "
                         "launcher's WORLD_SIZE in the environment this is what happens anyway)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(nranks, child_cmd, ndev, out=None, poll_s=0.2):
    """One command, all GPUs (what train/main.py:869-874 gets from torch.nn.DataParallel): start
    `nranks` fresh child processes running `child_cmd`, one per device, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relay rank 0's standard output line by line to `out`,
    and return the first non-zero exit code of any child (0 if all succeed).  The caller must not
    have touched the GPU: the children are fresh processes (never an exec of this one), and this
    parent only waits.  A child that fails ends the others (by their exact PIDs)."""
    import subprocess
    import threading
    out = out or sys.stdout
    if nranks < 1:
        raise ValueError("need at least one rank")
    if nranks > ndev:
        print(f"bench.py: --gpus {nranks} but only {ndev} GPU(s) are visible; one device per rank is needed",
              file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks),
                    "LOCAL_WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port,
                    "ERFNET_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(list(child_cmd), env=env, stdout=subprocess.PIPE, text=True, bufsize=1))

    def relay(proc, sink):
        for line in proc.stdout:
            sink.write(line)
            sink.flush()
    threads = [threading.Thread(target=relay, args=(p, out if r == 0 else sys.stderr), daemon=True)
               for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    code = 0
    live = set(range(nranks))
    kill_at = None                      # a rank that ignores SIGTERM (stuck in a collective) is killed
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 128 - rc
                print(f"bench.py: rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
                kill_at = time.monotonic() + 20.0
        if live and kill_at is not None and time.monotonic() > kill_at:
            for q in live:
                procs[q].kill()
            kill_at = None
        if live:
            time.sleep(poll_s)
    for r, p in enumerate(procs):
        try:
            p.wait(timeout=30)
        except Exception:
            p.kill()
    for t in threads:
        t.join(timeout=10)
    return code


def build_model(name):
    if name == "net_classification":
        from erfnet_pytorch_amd import net_classification
        return net_classification.Net(softmax_classes=CLASSES)
    from erfnet_pytorch_amd import erfnet
    return erfnet.ERFNet(CLASSES)


def logits_of(out):
    return out[0] if isinstance(out, (tuple, list)) else out


def act_dtype(name):
    return torch.bfloat16 if name == "bf16" else torch.float32


def event_time_ms(fn, device, reps, warm=5):
    """Average milliseconds of fn() over reps calls, HIP events on the stream the kernels run on."""
    stream = torch.cuda.current_stream(device)
    for _ in range(warm):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(reps):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def step_share(kernel_prefix):
    """Share of the fp32 training step's kernel time spent in kernels whose name starts with
    `kernel_prefix`, from the committed rocprofv3 --kernel-trace --stats summary of the step."""
    import csv
    path = os.path.join(ROOT, "profiles", "r3_train_f32_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    try:
        rows = list(csv.DictReader(open(path)))
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        mine = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].replace("void ", "").startswith(kernel_prefix))
        return round(mine / total, 4) if total else None
    except Exception:
        return None


def line_conv_roofline(device, batch, c, div, dil):
    """One launch of line_x3r_kernel<c,64>: a 3-tap 1-D convolution c -> c channels ((3,1), dilation
    `dil`, + bias + ReLU: the first convolution of a non_bottleneck_1d) over batch x H/div x W/div pixels
    of fp32 NHWC tensors.  Every fp32 product is formed from three bf16 terms per operand (six bf16
    MFMAs per k-step, fp32 accumulation), so the matrix ceiling is the dense bf16 peak / 6; the memory
    floor is input read + output written once.  erf_conv_line_forward_split issues exactly one such
    launch per call (weights are packed once, outside the timed region)."""
    from erfnet_pytorch_amd import functional as hipf
    h, w_ = HEIGHT // div, WIDTH // div
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(batch, c, h, w_, generator=g).clamp_(min=0).to(device)
    x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 1, generator=g) * 0.05).to(device)
    b = torch.zeros(c, device=device)
    packed = hipf.conv_line_pack_split(w)
    ms = event_time_ms(lambda: hipf.conv_line_split(x, packed, b, 0, dil, True), device, 50)
    pixels = batch * h * w_
    flops = 2.0 * pixels * (3 * c) * c
    nbytes = 2 * pixels * c * 4
    t = ms * 1e-3
    mfma_peak = BF16_MATRIX_PEAK_TFLOPS / 6.0
    mfma_floor_us = flops / (mfma_peak * 1e12) * 1e6
    hbm_floor_us = nbytes / (HBM_PEAK_GBS * 1e9) * 1e6
    name = f"line_x3r_kernel<{c}, 64>"
    return {"kernel": f"{name} (conv3x1 dilation {dil} + bias + ReLU, {c}->{c} channels, {batch}x{h}x{w_} pixels, "
                      "fp32 tensors, split-bf16 arithmetic)",
            "us_per_launch": round(ms * 1e3, 2), "flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
            "tflops": round(flops / t / 1e12, 2), "mfma_peak_tflops": round(mfma_peak, 1),
            "mfma_frac": round(flops / t / 1e12 / mfma_peak, 4), "mfma_floor_us": round(mfma_floor_us, 2),
            "executed_bf16_tflops": round(6 * flops / t / 1e12, 1),
            "hbm_gbs": round(nbytes / t / 1e9, 1), "hbm_frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
            "hbm_floor_us": round(hbm_floor_us, 2),
            "share_of_step": step_share(name)}


def mfma_calibration():
    """What the matrix pipe sustains on this box: scripts/ubench/bf16_mfma_rate (built by
    __graft_entry__.build()) runs a bare v_mfma_f32_32x32x16_bf16 loop on every SIMD, once on constant and
    once on random operands.  The instruction takes 32 cycles either way, but with toggling operands the
    part holds a lower clock, so the data-sheet peak (2.4 GHz) is not what any kernel with real data can
    reach.  Reported beside `roofline`, never in place of `peak`.  None when the binary is missing."""
    exe = os.path.join(ROOT, "scripts", "ubench", "bf16_mfma_rate")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "calib"], capture_output=True, text=True, timeout=60)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:          # a diagnostic: never fails the bench line
        return {"error": str(e)[:200]}


def dominant_kernel_roofline(device, batch):
    """The fp32 training step spends half of its time in the two widths of line_x3r_kernel (every 1-D
    convolution of the 64- and 128-channel non_bottleneck_1d blocks, forward and data gradient).  Both
    are timed live with HIP events on the stream they run on; `roofline` is the width with the larger
    share of the step in the committed profile (profiles/r3_train_f32_kernel_stats.csv), the other one
    is reported beside it.  The bound is whichever floor is higher for that launch: the bf16 matrix
    pipe at six MFMA terms per fp32 product (dense bf16 peak / 6), or HBM at input + output once."""
    k64 = line_conv_roofline(device, batch, 64, 4, 1)
    k128 = line_conv_roofline(device, batch, 128, 8, 2)
    s64, s128 = k64["share_of_step"] or 0.0, k128["share_of_step"] or 0.0
    dom, other = (k64, k128) if s64 >= s128 else (k128, k64)
    # HBM bytes per launch from the committed PMC passes (same launch geometry), per width
    tj = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    source = None
    try:
        recs = json.load(open(tj)).get("kernels", {})
    except Exception:
        recs = {}
    for k in (k64, k128):
        rec = recs.get(k["kernel"].split(" (")[0])
        k["hbm_bytes_per_launch_pmc"] = rec.get("hbm_bytes_per_launch") if rec else None
    traffic = dom["hbm_bytes_per_launch_pmc"]
    if traffic is not None:
        source = "profiles/dominant_kernel_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " \
                 "not measured in this run)"
    hbm_bound = dom["hbm_floor_us"] >= dom["mfma_floor_us"]
    out = {"bound": "hbm" if hbm_bound else "mfma",
           "achieved": dom["hbm_gbs"] if hbm_bound else dom["tflops"],
           "peak": HBM_PEAK_GBS if hbm_bound else dom["mfma_peak_tflops"],
           "unit": "GB/s" if hbm_bound else "TFLOP/s",
           "frac": dom["hbm_frac"] if hbm_bound else dom["mfma_frac"],
           "traffic": traffic, "traffic_source": source,
           "peak_note": "mfma: dense bf16 MFMA peak / 6 (each fp32 product costs six bf16 MFMA terms; the "
                        "convolution's algorithmic FLOPs are counted once); hbm: 8 TB/s against input read + "
                        "output written once",
           "frac_of_fp32_matrix_peak": round(dom["tflops"] / FP32_MATRIX_PEAK_TFLOPS, 4)}
    out.update(dom)
    out["other_width"] = other
    return out


def dominant_kernel_roofline_bf16(device, batch):
    """--dtype bf16: line2_bf16_kernel<128,64> — two chained 1-D convolutions 128 -> 128 channels over
    batch x 64 x 128 pixels in one launch (conv1x3 then conv3x1, intermediate stored as in a training
    forward), bf16 operands, fp32 accumulation: the kernel the bf16 step spends most of its time in.
    erf_conv_pair_forward_bf16 issues exactly one launch per call (weights packed once)."""
    import ctypes as C
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(batch, DOM_C, DOM_H, DOM_W, generator=g).clamp_(min=0).to(device).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    wa = (torch.randn(DOM_C, DOM_C, 3, generator=g) * 0.05).to(device)
    wb = (torch.randn(DOM_C, DOM_C, 3, generator=g) * 0.05).to(device)
    b = torch.zeros(DOM_C, device=device)
    packed = torch.empty(2 * 3 * DOM_C * DOM_C, dtype=torch.bfloat16, device=device)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.erf_conv_pair_pack_bf16(ptr(wa), ptr(wb), DOM_C, ptr(packed), st))
    y, mid = torch.empty_like(x), torch.empty_like(x)

    def call():
        _lib.check(lib.erf_conv_pair_forward_bf16(ptr(x), ptr(packed), ptr(b), ptr(b), 1, 2, 2, 1, 1, None, ptr(mid),
                                                  ptr(y), batch, DOM_C, DOM_H, DOM_W, st))
    ms = event_time_ms(call, device, 50)
    pixels = batch * DOM_H * DOM_W
    flops = 2 * 2.0 * pixels * (3 * DOM_C) * DOM_C
    nbytes = 3 * pixels * DOM_C * 2
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
            "kernel": "line2_bf16_kernel<128,64> (conv1x3 + conv3x1, 128->128 channels, one launch, bf16 tensors)",
            "us_per_launch": round(ms * 1e3, 2), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": nbytes,
            "hbm_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def harness_loop(call, device, batch, iters):
    """The loop of eval/eval_forwardTime.py:40-55: device synchronised around every call, the first
    call excluded, time per image = t / B (the script's sleep(1) dropped)."""
    times = []
    for i in range(iters + 1):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize(device)
        if i:
            times.append(time.perf_counter() - t0)
    return round(sum(times) / len(times) / batch * 1e3, 4), round(min(times) / batch * 1e3, 4)


def forward_time_harness(device, height, width, dtype_name, batch=1, iters=20):
    """The reference's forward-time script (eval/eval_forwardTime.py:22-58): erfnet_nobn.ERFNet(19)
    in eval mode on a fixed torch.randn(B,3,H,W); eager, and replayed from one HIP graph."""
    from erfnet_pytorch_amd import erfnet_nobn
    from erfnet_pytorch_amd.graphs import GraphedForward
    model = erfnet_nobn.ERFNet(19).to(device).eval()
    images = torch.randn(batch, 3, height, width, generator=torch.Generator().manual_seed(0)).to(device)
    images = images.to(act_dtype(dtype_name))
    out = {"model": "erfnet_nobn.ERFNet(19)", "batch": batch, "input": f"randn 3x{height}x{width}, {dtype_name}"}
    with torch.no_grad():
        out["ms_per_img_mean"], out["ms_per_img_min"] = harness_loop(lambda: model(images), device, batch, iters)
    graphed = GraphedForward(model, images)
    out["hip_graph_ms_per_img_mean"], out["hip_graph_ms_per_img_min"] = \
        harness_loop(lambda: graphed(images), device, batch, iters)
    gflop = FWD_GFLOP_PER_IMAGE * (height * width) / (HEIGHT * WIDTH)
    out["hip_graph_tflops"] = round(gflop / out["hip_graph_ms_per_img_min"], 2)
    return out


def nb1d_bandwidth(device, height, width, dtype_name):
    """Bandwidth of the non_bottleneck_1d blocks alone, as the encoder of a height x width image
    runs them: the five 64-channel blocks at 1/4 resolution and the eight 128-channel blocks at 1/8
    (dilations 2,4,8,16 twice), inference, each stage as one erf_nb1d_chain_forward call, timed with
    HIP events.  Algorithmic bytes: every block reads its input once and writes its output once."""
    from erfnet_pytorch_amd import erfnet_blocks as eb
    dt = act_dtype(dtype_name)
    esz = 2 if dtype_name == "bf16" else 4
    res = {}
    for c, div, dils in ((64, 4, [1] * 5), (128, 8, [2, 4, 8, 16] * 2)):
        layers = torch.nn.ModuleList([eb.non_bottleneck_1d(c, 0.0, d) for d in dils]).to(device).eval()
        x = torch.randn(1, c, height // div, width // div, device=device).clamp_(min=0).to(dt)
        x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            ms = event_time_ms(lambda: eb.run_layers(layers, x), device, 30)
        px = (height // div) * (width // div)
        nbytes = len(dils) * 2 * px * c * esz
        flops = len(dils) * 4 * 2.0 * px * 3 * c * c
        res[f"c{c}"] = {"blocks": len(dils), "pixels": px, "us_per_block": round(ms * 1e3 / len(dils), 2),
                        "algorithmic_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                        "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "tflops": round(flops / (ms * 1e-3) / 1e12, 1)}
    return res


def train_rate(device, model_name, dtype_name, batch, steps, warmup):
    """images/s of the training step at one GPU, built like the headline step: extra keys only."""
    from erfnet_pytorch_amd import optim
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    torch.manual_seed(0)
    from erfnet_pytorch_amd.ddp import GradientReducer
    model = build_model(model_name).to(device).train()
    crit = CrossEntropyLoss2d()
    g = torch.Generator(device="cpu").manual_seed(1000)
    images = torch.rand(batch, 3, HEIGHT, WIDTH, generator=g).to(device).to(act_dtype(dtype_name))
    labels = torch.randint(-1, CLASSES, (batch, HEIGHT, WIDTH), generator=g).to(device)
    # the headline step's set-up: gradients written into flat buckets, parameters flattened beside them,
    # the library's Adam on the flat buffers
    reducer = GradientReducer(model, bucket_bytes=3_600_000, direct_write=True)
    reducer.zero_grad()
    crit(logits_of(model(images)), labels).backward()
    reducer.finish()
    opt = optim.Adam(reducer.flatten_parameters(), 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)

    def step():
        reducer.zero_grad()
        loss = crit(logits_of(model(images)), labels)
        loss.backward()
        reducer.finish()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"images_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": batch,
            "model": model_name, "dtype": dtype_name, "final_loss": round(float(loss.item()), 5)}


def imagenet_rate(device, dtype_name, batch=256, steps=5, warmup=2):
    """The ImageNet pre-training step of the encoder (imagenet/erfnet_imagenet.py, imagenet/main.py:
    83-85,168-189): ERFNet(1000) with BatchNorm in train mode and Dropout2d, CrossEntropyLoss,
    SGD(lr 0.1, momentum 0.9, weight decay 1e-4), synthetic 256 x 3 x 224 x 224 batch per GPU."""
    from erfnet_pytorch_amd import erfnet_imagenet, optim
    torch.manual_seed(0)
    model = erfnet_imagenet.ERFNet(1000).to(device).train()
    opt = optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss()
    images = torch.randn(batch, 3, 224, 224, device=device).to(act_dtype(dtype_name))
    target = torch.randint(0, 1000, (batch,), device=device)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(images).float(), target)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"images_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "batch": batch,
            "dtype": dtype_name, "model": "erfnet_imagenet.ERFNet(1000), BatchNorm train mode, SGD"}


def cpu_baseline():
    """What the reference runs on a CPU — the same modules evaluated with stock torch CPU ops
    (forward + CrossEntropy + backward of 512x1024 images, no optimiser step) — timed on this host's
    cores; the plain-C oracle's time for the same pass is given beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    model = build_model("net_classification")
    rng = np.random.default_rng(0)
    x = rng.random((1, 3, HEIGHT, WIDTH), dtype=np.float32)
    lab = rng.integers(-1, CLASSES, (1, HEIGHT, WIDTH))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    model.train()
    torch.set_num_threads(cores)
    xt, lt = torch.from_numpy(x), torch.from_numpy(lab)
    from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d
    crit = CrossEntropyLoss2d()       # train/main.py:186-188
    with torch.no_grad():
        model(xt)
        t0 = time.perf_counter()
        for _ in range(3):
            model(xt)
        tf = (time.perf_counter() - t0) / 3
    crit(logits_of(model(xt)), lt).backward()
    n = 0
    t0 = time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 40):
        model.zero_grad(set_to_none=True)
        crit(logits_of(model(xt)), lt).backward()
        n += 1
    tt = (time.perf_counter() - t0) / n
    out = {"value": round(1.0 / tt, 3), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "%d passes over one 512x1024 image: forward + CrossEntropy + backward of "
                     "net_classification.Net(20) through stock torch CPU ops on the drop-in modules "
                     "(the arithmetic the reference runs on a CPU), no optimiser step; forward alone "
                     "%.0f ms/img" % (n, tf * 1e3)}
    try:
        from oracle import oracle as orc
        state = {k: v.detach().numpy() for k, v in model.state_dict().items()}
        t0 = time.perf_counter()
        orc.erfnet_train_pass(x, lab, state, "train")
        out["c_oracle_images_per_s"] = round(1.0 / (time.perf_counter() - t0), 4)
    except Exception as e:                                  # the oracle is the checker, not the baseline
        out["c_oracle_images_per_s"] = None
        out["c_oracle_error"] = str(e)[:200]
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # No launcher above us: be the launcher.  Nothing here initialises the GPU
        # (torch.cuda.device_count() only counts), the ranks are fresh child processes.
        child = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
        sys.exit(launch_ranks(args.gpus, child, torch.cuda.device_count()))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ERFNET_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("bench.py needs a GPU (the HIP kernels are the only compute path for the measured step)")
    if world > 1 and backend == "nccl" and (world > ndev or local_rank >= ndev):
        # two ranks sharing one device hang inside RCCL's first collective: refuse instead
        sys.exit(f"bench.py: WORLD_SIZE {world} (LOCAL_RANK {local_rank}) but only {ndev} GPU(s) are visible; "
                 f"RCCL needs one device per rank")
    if world > 1:
        # A rank stuck in a collective would otherwise sit silent until the launcher's own limit: dump
        # every thread's stack and exit instead (ERFNET_BENCH_WATCHDOG_S, default 900 s; 0 disables).
        import faulthandler
        limit = float(os.environ.get("ERFNET_BENCH_WATCHDOG_S", "900"))
        if limit > 0:
            faulthandler.dump_traceback_later(limit, exit=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE is {world}; using {world}", file=sys.stderr)
    # the matrix-pipe calibration is a child program: it runs before this process has touched the GPU
    cal = mfma_calibration() if world == 1 and not args.no_calibration else None
    local_rank %= ndev                  # (only reachable with a non-RCCL rehearsal backend)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from erfnet_pytorch_amd import _lib, optim
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
    images = torch.rand(args.batch, 3, HEIGHT, WIDTH, generator=g).to(device).to(act_dtype(args.dtype))
    labels = torch.randint(-1, CLASSES, (args.batch, HEIGHT, WIDTH), generator=g).to(device)

    # Set-up pass (no update): one backward tells the reducer which parameters take part; they are
    # then laid out in flat buffers next to their gradient buckets.  The optimiser keeps the
    # reference's settings (train/main.py:43-46,213); on the flat buffers the library's multi-tensor
    # Adam (erf_adam_step) is the same element-wise update in one launch for the whole model, and
    # parameters that never get a gradient are skipped as Adam skips them.
    reducer.zero_grad()
    crit(logits_of(model(images)), labels).backward()
    reducer.finish()
    params = reducer.flatten_parameters()
    opt = optim.Adam(params, 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6)

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

    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        ms_step = elapsed / args.steps * 1e3
        esz = 0.5 if args.dtype == "bf16" else 1.0
        step_gflop = 3 * FWD_GFLOP_PER_IMAGE * args.batch
        step_mb = 3 * FWD_MB_PER_IMAGE_F32 * esz * args.batch
        gpu = torch.cuda.get_device_name(device)
        bn = "BatchNorm off" if args.model == "net_classification" else "BatchNorm (train mode) + Dropout2d"
        rec = {
            "metric": "images/sec train (512x1024, 20cls)",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (uniform [0,1) images, labels in [-1,20), random-init weights)",
            "config": {"workload": f"{world}x{gpu} training, batch={args.batch} per GPU, {HEIGHT}x{WIDTH}, "
                                   f"{CLASSES} classes, {args.dtype}, encoder+decoder ({args.model}, {bn})"
                                   + (" [BASELINE configs[1]]" if (world == 1 and args.batch == 6 and
                                                                  args.dtype == "f32" and
                                                                  args.model == "net_classification") else ""),
                       "model": args.model, "per_gpu_batch": args.batch,
                       "global_batch": world * args.batch,
                       "step": "forward + CrossEntropy(ignore -1) + backward + Adam(lr 1e-4, eps 1e-4, wd 1e-6)",
                       "parallelism": f"dp{world}",
                       "arithmetic": ("bf16 operands, fp32 accumulation" if args.dtype == "bf16" else
                                      "fp32 tensors and fp32 accumulation; the 64/128-channel convolutions and "
                                      "weight gradients form each fp32 product from three bf16 terms per operand "
                                      "on the bf16 matrix pipe (error against float64 equal to the fp32-MFMA "
                                      "kernels', tests/test_split_bf16.py; ERFNET_NO_X3=1 selects those)"),
                       "grad_buckets_bytes": reducer.bucket_sizes},
            "final_loss": round(final_loss, 5),
            # whole step against both rooflines (algorithmic work of the blocks / measured step time)
            "step_tflops": round(step_gflop / ms_step, 2),
            "step_algorithmic_gbs": round(step_mb / ms_step, 1),
            "step_hbm_frac": round(step_mb / ms_step / HBM_PEAK_GBS, 4),
        }
        if world == 1:
            del model, opt, params, reducer
            torch.cuda.empty_cache()
        # the dominant kernel is timed on rank 0's device at every world size (one launch geometry per
        # GPU whatever N is: the scaling is weak); the other ranks wait in the closing barrier meanwhile
        rec["roofline"] = (dominant_kernel_roofline_bf16(device, args.batch) if args.dtype == "bf16" else
                           dominant_kernel_roofline(device, args.batch))
        if world == 1:
            if cal and "random_operands_tflops" in cal:
                r = rec["roofline"]
                # bf16 MFMA work the timed kernel executes: six terms per fp32 product, one per bf16 product
                executed = (r["achieved"] if args.dtype == "bf16" else 6.0 * r["tflops"])
                cal["kernel_bf16_mfma_tflops"] = round(executed, 1)
                cal["kernel_frac_of_random_operands_rate"] = round(executed / cal["random_operands_tflops"], 4)
            rec["roofline"]["mfma_calibration"] = cal
            if not args.no_extras:
                other = "bf16" if args.dtype == "f32" else "f32"
                rec["train_other_dtype"] = train_rate(device, args.model, other, args.batch, 30, 8)
                rec["forward_time_harness"] = forward_time_harness(device, HEIGHT, WIDTH, "f32")
                # BASELINE configs[2]: eval_forwardTime.py path, erfnet_nobn, bf16, 1024x2048
                full = forward_time_harness(device, 2 * HEIGHT, 2 * WIDTH, "bf16")
                full["nb1d"] = nb1d_bandwidth(device, 2 * HEIGHT, 2 * WIDTH, "bf16")
                full["f32_ms_per_img_min"] = forward_time_harness(device, 2 * HEIGHT, 2 * WIDTH, "f32",
                                                                  iters=10)["ms_per_img_min"]
                rec["fullres_inference_bf16"] = full
                rec["train_erfnet_bn"] = train_rate(device, "erfnet_bn", args.dtype, args.batch, 20, 6)
                rec["train_erfnet_bn_other_dtype"] = train_rate(device, "erfnet_bn", "bf16" if args.dtype == "f32" else "f32",
                                                                args.batch, 20, 6)
                # BASELINE configs[4] per GPU: the ImageNet pre-training step of the encoder
                rec["imagenet_pretrain"] = {"f32": imagenet_rate(device, "f32"),
                                            "bf16": imagenet_rate(device, "bf16")}
            if not args.no_cpu_baseline:
                rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
