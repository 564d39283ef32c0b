"""bench.py's training step: host enqueue time, eager step time, and the same step captured once in a
HIP graph and replayed (loss trajectory compared with the eager one)."""
import os, sys, time, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from erfnet_pytorch_amd.ddp import GradientReducer
from erfnet_pytorch_amd.evaluation_functions import CrossEntropyLoss2d

dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"

def setup():
    torch.manual_seed(0)
    model = bench.build_model("net_classification").to(dev).train()
    red = GradientReducer(model, bucket_bytes=3_600_000, direct_write=True)
    crit = CrossEntropyLoss2d()
    g = torch.Generator(device="cpu").manual_seed(1000)
    images = torch.rand(6, 3, 512, 1024, generator=g).to(dev).to(bench.act_dtype(dtype))
    labels = torch.randint(-1, 20, (6, 512, 1024), generator=g).to(dev)
    red.zero_grad(); crit(bench.logits_of(model(images)), labels).backward(); red.finish()
    params = red.flatten_parameters()
    return model, red, crit, images, labels, params

model, red, crit, images, labels, params = setup()
opt = torch.optim.Adam(params, 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6, fused=True)
def step():
    red.zero_grad()
    loss = crit(bench.logits_of(model(images)), labels)
    loss.backward(); red.finish(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
host, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); tot.append(t2 - t0)
print(f"eager: host enqueue {min(host)*1e3:.2f} ms, step {min(tot)*1e3:.2f} ms")
eager_losses = [float(step()) for _ in range(4)]

# the same from scratch, captured
model, red, crit, images, labels, params = setup()
opt = torch.optim.Adam(params, 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6, fused=True, capturable=True)
def step2():
    red.zero_grad()
    loss = crit(bench.logits_of(model(images)), labels)
    loss.backward(); red.finish(); opt.step()
    return loss
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step2()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_loss = step2()
torch.cuda.synchronize()
for _ in range(2): graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): graph.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")
# the captured run did 3 warm-up steps + 1 capture pass (not executed) + 22 replays; compare the first replays
model, red, crit, images, labels, params = setup()
opt = torch.optim.Adam(params, 1e-4, (0.9, 0.999), eps=1e-4, weight_decay=1e-6, fused=True, capturable=True)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step2()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_loss = step2()
gl = []
for _ in range(4):
    graph.replay(); gl.append(float(static_loss))
print("eager losses (steps 4-7):", eager_losses)
print("graph losses (steps 4-7):", gl)
