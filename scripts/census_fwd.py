"""Census of the GPU launches of one full-resolution bf16 inference forward (erfnet_nobn.ERFNet(19), 1x3x1024x2048)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet_nobn
from torch.profiler import ProfilerActivity, profile
net = erfnet_nobn.ERFNet(19).to("cuda:0").eval()
x = torch.randn(1, 3, 1024, 2048, device="cuda:0").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(2): net(x)
        torch.cuda.synchronize()
groups = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"): continue
    if e.name not in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::empty_like"): continue
    frame = ""
    for f in (e.stack or []):
        if "erfnet_pytorch_amd" in f:
            frame = f.split("/")[-1]; break
    groups[(e.name, str(e.input_shapes)[:70], frame[:100])] += 1
for (name, shapes, frame), n in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"{n/2:6.1f}  {name:24s} {shapes:70s} {frame}")
