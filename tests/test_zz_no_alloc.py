"""erf_cross_entropy_forward_static — the entry point that allocates nothing (its block partial sums
live in a device array of the library's code object) — against erf_cross_entropy_forward with a
caller workspace, through the raw C ABI: bit-identical loss, pixel count and per-pixel log-sum-exp,
on two calls in a row, the second with a smaller input (the shared array keeps the first call's
contents), and a clear rejection beyond the size limit."""
import ctypes as C
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _both(n, k, h, w, tag):
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    logits = torch.as_tensor(gu.det_normal(tag + ".logits", (n, k, h, w))).to(DEV) * 3.0
    target = torch.as_tensor(gu.det_labels(tag + ".lab", (n, h, w), -1, k)).to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for static in (False, True):
        stats = torch.full((2,), -7.0, device=DEV)
        lse = torch.full((n, h, w), -7.0, device=DEV)
        if static:
            rc = lib.erf_cross_entropy_forward_static(_ptr(logits), _ptr(target), _ptr(stats), _ptr(lse),
                                                      n, k, h, w, -1, st)
        else:
            ws = torch.empty(int(lib.erf_workspace_bytes(n * h * w, k)), dtype=torch.uint8, device=DEV)
            rc = lib.erf_cross_entropy_forward(_ptr(logits), _ptr(target), _ptr(stats), _ptr(lse),
                                               n, k, h, w, -1, _ptr(ws), ws.numel(), st)
        _lib.check(rc)
        torch.cuda.synchronize()
        out.append((stats.cpu(), lse.cpu()))
    # the checker beside it: torch's own loss on the same data
    ref = torch.nn.functional.cross_entropy(logits.double(), target, ignore_index=-1).item()
    return out, ref


@pytest.mark.timeout(120)
def test_static_scratch_cross_entropy_matches_workspace_entry_point():
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    before = _lib.launch_count()
    # first call: 3 x 20 x 96 x 160 (180 blocks); second, smaller: 1 x 20 x 40 x 56 (9 blocks, ragged tail)
    for shape, tag in (((3, 20, 96, 160), "noalloc.a"), ((1, 20, 40, 56), "noalloc.b"), ((2, 40, 33, 47), "noalloc.c")):
        (ws_stats, ws_lse), (st_stats, st_lse) = _both(*shape, tag)[0]
        assert torch.equal(ws_stats, st_stats), (shape, ws_stats, st_stats)
        assert torch.equal(ws_lse, st_lse), shape
        ref = _both(*shape, tag)[1]
        assert abs(st_stats[0].item() - ref) <= 1e-5 * max(1.0, abs(ref)), (st_stats, ref)
    assert _lib.launch_count() > before
    # beyond the limit: rejected before any launch, with a message
    n_launch = _lib.launch_count()
    rc = lib.erf_cross_entropy_forward_static(None, None, None, None, 64, 20, 1024, 1024, -1, None)
    assert rc != 0
    assert b"exceed the limit" in lib.erf_last_error()
    assert _lib.launch_count() == n_launch


@pytest.mark.timeout(120)
def test_static_scratch_cross_entropy_replays_from_a_hip_graph():
    """The callers the entry point exists for: after one ordinary call on the device (which resolves
    the scratch array's address), the call is captured into a HIP graph and replayed on new data —
    nothing in it allocates, and the replayed results equal the workspace entry point's."""
    from erfnet_pytorch_amd import _lib
    lib = _lib.lib()
    n, k, h, w = 2, 20, 48, 64
    logits = torch.as_tensor(gu.det_normal("noalloc.g.logits", (n, k, h, w))).to(DEV) * 2.0
    target = torch.as_tensor(gu.det_labels("noalloc.g.lab", (n, h, w), -1, k)).to(DEV)
    stats = torch.zeros(2, device=DEV)
    lse = torch.zeros((n, h, w), device=DEV)

    def call(stream):
        return lib.erf_cross_entropy_forward_static(_ptr(logits), _ptr(target), _ptr(stats), _ptr(lse), n, k, h, w, -1,
                                                    C.c_void_p(stream.cuda_stream))
    _lib.check(call(torch.cuda.current_stream()))          # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph, stream=side):
        rc = call(torch.cuda.current_stream())
    _lib.check(rc)
    for seed in (1, 2):
        logits.copy_(torch.as_tensor(gu.det_normal(f"noalloc.g.logits{seed}", (n, k, h, w))).to(DEV) * 2.0)
        stats.fill_(-1.0)
        lse.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        ws = torch.empty(int(lib.erf_workspace_bytes(n * h * w, k)), dtype=torch.uint8, device=DEV)
        s2, l2 = torch.zeros(2, device=DEV), torch.zeros((n, h, w), device=DEV)
        _lib.check(lib.erf_cross_entropy_forward(_ptr(logits), _ptr(target), _ptr(s2), _ptr(l2), n, k, h, w, -1,
                                                 _ptr(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.equal(stats, s2) and torch.equal(lse, l2), seed
