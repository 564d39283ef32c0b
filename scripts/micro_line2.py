"""One shape of the bf16 pair kernel (erf_conv_pair_forward_bf16: conv1x3 then conv3x1, intermediate
stored as in a training forward) for rocprofv3 kernel-trace / PMC passes: `micro_line2.py C H W DIL [reps]`."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from erfnet_pytorch_amd import _lib

lib = _lib.lib()
DEV = "cuda:0"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
c, h, w, dil = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
n = 6
x = torch.randn(n, c, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w1 = torch.randn(c, c, 3, device=DEV) * 0.05
w2 = torch.randn(c, c, 3, device=DEV) * 0.05
b = torch.zeros(c, device=DEV)
packed = torch.empty(2 * 3 * c * c, dtype=torch.bfloat16, device=DEV)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.erf_conv_pair_pack_bf16(p(w1), p(w2), c, p(packed), st))
y, m = torch.empty_like(x), torch.empty_like(x)
for _ in range(reps + 3):
    _lib.check(lib.erf_conv_pair_forward_bf16(p(x), p(packed), p(b), p(b), 1, dil, dil, 1, 1, None, p(m), p(y), n, c, h, w, st))
torch.cuda.synchronize()
print("done", c, h, w, dil)
