"""erfnet_nobn.ERFNet(19) inference at 1024x2048, eager: `fwd_fullres.py [batch] [f32|bf16]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from erfnet_pytorch_amd import erfnet_nobn
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = erfnet_nobn.ERFNet(19).to("cuda:0").eval()
dt_ = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
x = torch.randn(b, 3, 1024, 2048, device="cuda:0").to(dt_)
with torch.no_grad():
    for _ in range(3):
        y = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        y = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"1024x2048 batch {b}: {dt*1e3/b:.3f} ms/img  pair={os.environ.get('ERFNET_NO_PAIR','0')!='1'}  finite={bool(torch.isfinite(y.float()).all())} {dt_}")
