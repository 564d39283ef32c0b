"""HIP-graph replay of an inference forward.

At batch 1 a 512x1024 forward is ~75 kernel launches of 10-60 us each: the host (Python, ctypes,
the launch path) cannot enqueue them as fast as the GPU retires them, so the eager forward is
launch-bound.  The library never allocates or synchronises and always launches on the current
stream, so the whole forward can be captured once into a HIP graph and replayed with a single
launch.  No counterpart in the reference (its scripts call the module eagerly); the captured
callable takes the place of ``model(images)`` in a serving loop or in the forward-time harness
(eval/eval_forwardTime.py:43-46).
"""
import torch


class GraphedForward:
    """``GraphedForward(model, example)(x)`` == ``model(x)`` for inputs of example's shape/dtype.

    The module must be in eval mode and is run under ``torch.no_grad()``; its parameters are read
    at replay time (updating them in place is seen by later replays, rebinding ``.data`` is not).
    The returned tensors are the graph's static outputs: they are overwritten by the next call.
    """

    def __init__(self, model, example, warmup=2):
        if model.training:
            raise RuntimeError("GraphedForward captures an inference forward: call model.eval() first")
        if not example.is_cuda:
            raise RuntimeError("GraphedForward needs a GPU tensor")
        self.model = model
        self.static_in = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            # the workspace cache is keyed by stream: warm up on the stream the capture runs on,
            # so the capture finds its workspace and allocates nothing
            for _ in range(warmup):
                model(self.static_in)
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=side):
            self.static_out = model(self.static_in)

    def __call__(self, x):
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"graph was captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
