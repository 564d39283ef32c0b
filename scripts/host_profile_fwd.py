"""cProfile of the host side of the batch-1 inference forward (erfnet_nobn.ERFNet(19), 512x1024)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import erfnet_nobn
m = erfnet_nobn.ERFNet(19).to("cuda:0").eval()
x = torch.randn(1, 3, 512, 1024, device="cuda:0")
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {(t1-t0)/50*1e3:.3f} ms / forward, with sync {(t2-t0)/50*1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        m(x)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
