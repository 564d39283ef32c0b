"""One-process-per-GPU data parallelism for the ERFNet training step.

The reference trains under single-process ``torch.nn.DataParallel`` (train/main.py:870-874): every
step it re-broadcasts the parameters from GPU 0, scatters the batch, gathers the outputs and
reduce-adds the gradients back onto GPU 0.  Here each rank owns one GPU and a full replica; the only
exchange is one sum of the gradients per step, issued in a few flat buckets from autograd hooks so
that the RCCL all-reduce of the decoder's gradients runs while the encoder is still in backward.

Design points
  * Gradients live in flat bucket buffers; each ``param.grad`` is a view into its bucket, so the
    collective runs in place on what autograd accumulated and nothing is copied.
  * Buckets are filled in reverse registration order (the order backward produces gradients).
  * Parameters that never receive a gradient (the reference's unused ``bn.*`` when BatchNorm is off,
    ``encoder.output_conv.*``, ``class_power``) are found on the first backward pass and left out of
    later steps instead of forcing ``find_unused_parameters``.
  * BatchNorm statistics stay per replica and dropout masks per rank, as under DataParallel.
  * With ``direct_write`` the block kernels write each weight gradient straight into its bucket
    slice (functional._grad_targets): no temporary, no accumulation kernel per parameter.  A
    parameter's buffer is granted once per step, so a second gradient of the same parameter
    (weight sharing) still accumulates through autograd; gradients therefore do NOT accumulate
    across several backward passes without ``zero_grad`` in between when this is on.
  * A step with more than one ``backward()`` (the reference's mean-teacher loop,
    train/main.py:377-379: ``loss_consistency.backward(retain_graph=True)`` followed by
    ``loss_student_pred.backward()``) or with shared weights must start with
    ``zero_grad(defer=True)``: the hooks then only record which parameters took part and
    ``finish()`` launches every bucket after the last pass.  Without it a second gradient arriving
    for a bucket whose all-reduce is already in flight raises instead of being lost.
  * ``zero_grad()`` of the reducer (not the optimiser's, whose ``set_to_none`` unbinds
    ``param.grad`` from the buckets) starts every step; ``finish()`` checks the binding.
  * The payload is small (8.2 MB of fp32 for the 20-class network), i.e. latency-bound on xGMI:
    few, large buckets beat many small ones.  Only the last bucket's all-reduce cannot overlap with
    backward, so the bucket size should leave a small remainder for it (the 128-channel stage,
    whose gradients come first, holds 6.3 of the 8.2 MB).
"""
import torch
import torch.distributed as dist


class GradientReducer:
    """Bucketed, overlapped gradient averaging for a replicated module."""

    def __init__(self, module, bucket_bytes=4 << 20, process_group=None, direct_write=False,
                 reduce_single_rank=False):
        self.module = module
        # a one-rank group normally skips the collectives; tests turn them on to drive the RCCL
        # path (enqueue order against the compute stream, in-place buckets) on a single GPU
        self._always = bool(reduce_single_rank)
        self._direct = bool(direct_write)
        self._written = set()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # RCCL averages inside the collective (ncclAvg); other backends sum and divide afterwards
        self._avg_op = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._bucket_bytes = bucket_bytes
        self._active = None            # params known to receive gradients (after the first step)
        self._buckets = []             # (flat buffer, [params])
        self._pending = {}             # bucket index -> number of params still missing this step
        self._bucket_of = {}
        self._work = []
        self._seen = set()
        self._defer = False
        self._grad_ptr = {}            # param -> data_ptr of its bucket slice
        self._first_step_done = False
        self._hooks = []
        self._build_buckets(self.params)
        self._pending = {i: len(ps) for i, (_, ps) in enumerate(self._buckets)}
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---------------------------------------------------------------------------------- set-up
    def broadcast_parameters(self, src=0):
        """Make every replica start from rank ``src``'s parameters and buffers."""
        if self.world == 1 and not self._always:
            return
        # one flat buffer per dtype: a few large broadcasts instead of several hundred small ones
        with torch.no_grad():
            by_dtype = {}
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                by_dtype.setdefault(t.dtype, []).append(t.data)
            for tensors in by_dtype.values():
                flat = torch.cat([t.reshape(-1) for t in tensors])
                dist.broadcast(flat, src, group=self.group)
                off = 0
                for t in tensors:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def _build_buckets(self, params):
        self._buckets, self._bucket_of, self._grad_ptr = [], {}, {}
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= self._bucket_bytes:
                self._finish_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._finish_bucket(cur)

    def _finish_bucket(self, params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            view = flat[off:off + p.numel()].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
            self._grad_ptr[p] = view.data_ptr()
            if self._direct:
                p._erf_grad_sink = (view, self)
            off += p.numel()
            self._bucket_of[p] = len(self._buckets)
        self._buckets.append((flat, list(params)))

    # ---------------------------------------------------------------------------------- per step
    def zero_grad(self, defer=False):
        """Zero the bucket buffers (keeps ``param.grad`` bound to them) and start a step.

        ``defer=True``: no all-reduce is launched from inside backward; ``finish()`` launches them all.
        Needed when the step calls ``backward()`` more than once or shares weights between layers."""
        self._defer = bool(defer)
        torch._foreach_zero_([flat for flat, _ in self._buckets])      # (one launch for all buckets)
        self._pending = {i: len(ps) for i, (_, ps) in enumerate(self._buckets)}
        self._work = []
        self._seen = set()
        self._written = set()

    # direct gradient write (see functional._grad_targets)
    def _claim(self, p):
        if not self._first_step_done or p in self._written or p not in self._bucket_of:
            return False
        self._written.add(p)
        return True

    def _delivered(self, p):
        # The kernel's write into the bucket slice has been enqueued.  Nothing is counted here: autograd
        # still runs the parameter's accumulate node (with an undefined gradient) and with it the
        # post-accumulate hook, which is where every gradient is counted once.  Should a torch version
        # skip the hook for undefined gradients, the bucket is simply launched from finish().
        pass

    def _on_grad(self, p):
        if self.world == 1 and self._first_step_done and not self._always:
            return
        b = self._bucket_of.get(p)
        if b is None:
            return
        if p in self._seen:
            # a further gradient of this parameter in the same step (second backward pass, shared
            # weights): fine while its bucket has not left yet, lost otherwise
            if self._pending.get(b, 0) < 0:
                raise RuntimeError("GradientReducer: a parameter received another gradient after its bucket's "
                                   "all-reduce was launched (several backward passes per step, or shared "
                                   "weights): start the step with zero_grad(defer=True)")
            return
        self._seen.add(p)
        self._pending[b] -= 1
        if self._pending[b] == 0 and self._first_step_done and not self._defer:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1 and not self._always:
            return
        flat, _ = self._buckets[b]
        op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
        self._work.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat))
        self._pending[b] = -1

    def finish(self):
        """Call after ``loss.backward()``: reduce what has not been launched, wait, average."""
        if not self._first_step_done:
            # first step: learn which parameters take part, then rebuild the buckets around them
            active = [p for p in self.params if p in self._seen]
            stale = [p for p in self.params if p not in self._seen]
            grads = {p: p.grad.clone() for p in active}
            for p in stale:
                p.grad = None
                if hasattr(p, "_erf_grad_sink"):
                    del p._erf_grad_sink
            for p in active:
                p.grad = grads[p]
            self._build_buckets(active)
            self._active = active
            self._first_step_done = True
            self._pending = {i: 0 for i in range(len(self._buckets))}
            for b in range(len(self._buckets)):
                self._launch(b)
        else:
            for p, ptr in self._grad_ptr.items():
                if p.grad is None or p.grad.data_ptr() != ptr:
                    raise RuntimeError("GradientReducer: a parameter's .grad no longer points into its bucket "
                                       "(optimizer.zero_grad(set_to_none=True)?): call reducer.zero_grad() "
                                       "before every backward pass instead")
            for b, left in self._pending.items():
                if left >= 0:          # bucket not launched (deferred, or some parameter got no gradient)
                    self._launch(b)
        for work, flat in self._work:
            work.wait()
            if not self._avg_op:
                flat.div_(self.world)
        self._work = []

    def flatten_parameters(self):
        """Move the parameters that receive gradients into flat buffers laid out like the gradient
        buckets and return one ``nn.Parameter`` per bucket whose ``.grad`` is that bucket.  An
        element-wise optimiser (Adam, SGD) built on the returned list performs exactly the update
        it would perform tensor by tensor — parameters without a gradient are skipped either way —
        in one launch per bucket instead of one pass over 150 small tensors.  Call after the first
        backward pass + ``finish()`` (the set of live parameters is known from then on); the
        module's parameters become views into the flat buffers."""
        if not self._first_step_done:
            raise RuntimeError("flatten_parameters() needs one backward pass and finish() first")
        flats = []
        with torch.no_grad():
            for flat_g, ps in self._buckets:
                flat_p = torch.empty_like(flat_g)
                off = 0
                for p in ps:
                    n = p.numel()
                    flat_p[off:off + n].copy_(p.data.reshape(-1))
                    p.data = flat_p[off:off + n].view_as(p)
                    off += n
                fp = torch.nn.Parameter(flat_p, requires_grad=True)
                fp.grad = flat_g
                flats.append(fp)
        self._flat_params = flats
        return flats

    @property
    def bucket_sizes(self):
        return [flat.numel() * flat.element_size() for flat, _ in self._buckets]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if hasattr(p, "_erf_grad_sink"):
                del p._erf_grad_sink
