"""Data-parallel training over the GPUs of one node: one process per GPU, sequences sharded across ranks,
gradients averaged with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference is single-GPU (no collective anywhere); this module ADDS data parallelism.
Design for xGMI (point-to-point links, per-link bound rings): all 68-77 gradient tensors live in ONE flat fp32 buffer
(59.5 MB for RAM-Net) whose views are the parameters' ``.grad``; it is all-reduced in a few large buckets on a side
HIP stream.  Every weight is used at every time step of the BPTT pass, so no gradient is final before the backward of time
step 0 has run; what CAN overlap is the end of the pass: the weight gradients accumulate in kernel-side workspaces that are
folded into ``.grad`` layer by layer when the autograd engine finishes (ops._Engine.flush: slab joins, Winograd-domain
un-transforms, the decoders' fold).  With ``overlap=True`` (default on GPUs) the reducer watches those folds and enqueues bucket
k on the side stream as soon as the last parameter of buckets 0..k is final — decoder / prediction layers first, then the
recurrent cells, then encoders and heads — so the collectives run under the remaining folds; all_reduce() after
backward() sends whatever is left (everything, when the pass had no kernel-side folds) and Adam waits on the event recorded
behind the last bucket.  Buckets are always issued in index order: every rank enqueues the same sequence of collectives.
Loss semantics: each rank's loss is the mean over ITS batch (standard DDP); gradients are averaged over ranks.
"""
import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, model, process_group=None, num_buckets=3, always_collective=False, overlap=True):
        """always_collective: issue the bucketed collectives even at world size 1 (an initialised process group is required) — the
        RCCL path of a single-GPU box is then the very code an 8-GPU run executes.  overlap: buckets leave during the end-of-backward
        fold (see the module text); False = everything in all_reduce()."""
        self.always_collective = bool(always_collective)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = process_group
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        # reverse registration order ~ order in which the backward pass finishes with the layers
        order = list(reversed(self.params))
        off, self.views = 0, {}
        for p in order:
            n = p.numel()
            self.views[p] = self.flat[off:off + n].view_as(p)
            off += n
        # bucket boundaries (in elements) at parameter boundaries, roughly equal sizes
        bounds, acc, target = [0], 0, total / float(max(1, num_buckets))
        for p in order:
            acc += p.numel()
            if acc >= target * len(bounds) and acc < total:
                bounds.append(acc)
        bounds.append(total)
        self.buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
        self.cuda = dev.type == "cuda"
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        self.done = None
        self.attach()
        # early issue during the fold: parameter -> bucket, buckets issued so far in this pass, folds still pending per bucket
        self._bucket_of, off = {}, 0
        for p in order:
            self._bucket_of[p] = next(i for i, (lo, hi) in enumerate(self.buckets) if lo <= off < hi)
            off += p.numel()
        self._issued, self._pending, self.early_buckets, self._armed = 0, None, 0, False
        self._hooked = False
        self.timing = False               # bench: bracket the collectives of a step with events on the side stream (span_ms())
        self._t0 = self._t1 = None
        self.issue_log = []               # (first bucket, one past the last) of every _issue() since the last zero(): the order buckets left in
        if overlap and self.cuda:
            from . import ops
            prev = ops.get_finalize_hook()
            if prev is not None and prev is not self:
                import warnings
                warnings.warn("FlatGradReducer: replacing the end-of-backward hook of another reducer (close() the old one first)")
            ops.set_finalize_hook(self)
            self._hooked = True

    def close(self):
        """Detach from the end-of-backward fold (the process-global ops hook keeps the reducer, its flat buffer and the model alive
        otherwise).  The hook is removed only if it still is this reducer; pending early buckets are waited for."""
        if self._hooked:
            from . import ops
            if ops.get_finalize_hook() is self:
                ops.set_finalize_hook(None)
            self._hooked = False
        if self.cuda and self._issued:
            torch.cuda.current_stream().wait_stream(self.side)
        self._issued, self._pending, self._armed = 0, None, False

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def attach(self):
        """(Re)install the flat-buffer views as .grad (call after zero_grad(set_to_none=True))."""
        for p, v in self.views.items():
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def zero(self, arm=True):
        """Start of a step: gradients to zero; arms the early issue for the ONE backward pass that follows.  A further pass before
        all_reduce() (gradient accumulation, two losses) is safe: the armed pass's early collectives are complete before anything later
        on the stream runs (end()), and begin() of the unarmed pass schedules every bucket for reduction again — the early buckets hold
        the rank AVERAGE of pass 1 on every rank, so averaging (that + the local gradients of pass 2) gives avg 1 + avg 2.  For
        accumulation loops prefer zero(arm=False) and arm() right before the last backward: the overlap then goes to the pass that
        completes the gradients.  arm=False: the pass that follows is NOT a collective one: nothing may leave early."""
        if self._issued:                  # buckets of a pass whose all_reduce() never came: let them finish before the buffer is reused
            torch.cuda.current_stream().wait_stream(self.side)
            self._issued, self._pending = 0, None
        self.flat.zero_()
        self.attach()
        self._armed = bool(arm)
        self.issue_log = []
        self._t0 = self._t1 = None

    def arm(self):
        """Arm the early issue for the NEXT backward pass (the last one of a gradient-accumulation step)."""
        self._armed = True

    def all_reduce(self):
        """Average gradients over ranks; asynchronous on the side stream (wait() before the optimizer)."""
        w = self.world
        for p, v in self.views.items():     # e.g. optimizer.zero_grad(set_to_none=True) after zero(): autograd allocated a
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():      # fresh .grad outside the flat buffer
                v.copy_(p.grad)
                p.grad = v
        if not self._collective():
            return
        if self.cuda:
            self._issue(len(self.buckets))
            if self.timing:
                self._t1 = torch.cuda.Event(enable_timing=True)
                self._t1.record(self.side)
            self.done = self.side.record_event()
            self._issued, self._pending = 0, None
        else:
            self.issue_log.append((0, len(self.buckets)))
            for lo, hi in self.buckets:
                chunk = self.flat[lo:hi]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                chunk.mul_(1.0 / w)

    def _collective(self):
        return self.world > 1 or (self.always_collective and dist.is_available() and dist.is_initialized())

    def _issue(self, upto):
        """Enqueue buckets [_issued, upto) on the side stream, behind everything the current stream has done so far."""
        if upto <= self._issued:
            return
        w = self.world
        self.side.wait_event(torch.cuda.current_stream().record_event())
        self.issue_log.append((self._issued, upto))
        with torch.cuda.stream(self.side):
            if self.timing and self._t0 is None:
                self._t0 = torch.cuda.Event(enable_timing=True)
                self._t0.record()
            for lo, hi in self.buckets[self._issued:upto]:
                chunk = self.flat[lo:hi]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                chunk.mul_(1.0 / w)
        self._issued = upto

    # ---- ops.set_finalize_hook protocol (called by ops._Engine.flush at the end of a backward pass)
    def begin(self, dirty):
        self._pending = None
        if not any(self._buckets_of(cp) for cp in dirty):      # a pass over another model: not this reducer's business
            return
        armed, self._armed = self._armed, False
        if self._issued:
            # a pass AFTER one that already sent buckets early (gradient accumulation with overlap): this pass's folds add local
            # gradients into buckets that hold rank averages — every bucket is reduced again by all_reduce() (linear: avg1 + avg2);
            # the collectives in flight are complete before the folds write (end() of the earlier pass made the stream wait)
            torch.cuda.current_stream().wait_stream(self.side)
            self._issued = 0
            return
        if not armed or not self._collective():
            return
        for p, v in self.views.items():             # a .grad outside the flat buffer (set_to_none between steps): all_reduce() repairs it
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                return
        self._pending = [0] * len(self.buckets)
        for cp in dirty:
            for b in self._buckets_of(cp):
                self._pending[b] += 1
        self._ready()

    def _buckets_of(self, cp):
        ps = list(cp.weights) + [b for b in cp.biases if b is not None]
        return {self._bucket_of[p] for p in ps if p in self._bucket_of}

    def finalized(self, cp):
        if self._pending is None:
            return
        for b in self._buckets_of(cp):
            self._pending[b] -= 1
        self._ready()

    def end(self):
        """End of the fold: everything enqueued on the stream afterwards (a second backward pass writing .grad directly — the
        prediction layer's backward does —, the optimizer) is ordered behind the early collectives.  The overlap that matters, bucket
        k's collective under the folds of the later layers, has happened by now."""
        if self._issued:
            torch.cuda.current_stream().wait_stream(self.side)

    def _ready(self):
        k = self._issued
        while k < len(self.buckets) and self._pending[k] == 0:
            k += 1
        self.early_buckets += k - self._issued
        self._issue(k)

    def span_ms(self):
        """timing=True: milliseconds on the side stream from the start of this step's first collective to the end of its last (after
        a device synchronisation); includes the time the early buckets waited for later folds."""
        if self._t0 is None or self._t1 is None:
            return None
        self._t1.synchronize()
        return self._t0.elapsed_time(self._t1)

    def wait(self):
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None


def shard_indices(n_items, rank, world):
    """Sequence sharding of SURVEY section 8e: rank r gets items r, r+world, ..."""
    return list(range(rank, n_items, world))
