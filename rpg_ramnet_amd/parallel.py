"""Data-parallel training over the GPUs of one node: one process per GPU, sequences sharded across ranks,
gradients averaged with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference is single-GPU (no collective anywhere); this module ADDS data parallelism.
Design for xGMI (point-to-point links, per-link bound rings): all 68-77 gradient tensors live in ONE flat fp32 buffer
(59.5 MB for RAM-Net) whose views are the parameters' ``.grad``; it is all-reduced in a few large buckets on a side
HIP stream.  Every weight is used at every time step of the BPTT pass, so no gradient is final before the pass ends
and its fold (ops._Engine.flush) has run: all_reduce() is therefore called AFTER backward() returns and overlaps only
host-side work (loss read-back, optimizer bookkeeping); Adam waits on the event recorded behind the last bucket.
Loss semantics: each rank's loss is the mean over ITS batch (standard DDP); gradients are averaged over ranks.
"""
import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, model, process_group=None, num_buckets=3, always_collective=False):
        """always_collective: issue the bucketed collectives even at world size 1 (an initialised process group is required) — the
        RCCL path of a single-GPU box is then the very code an 8-GPU run executes."""
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

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def attach(self):
        """(Re)install the flat-buffer views as .grad (call after zero_grad(set_to_none=True))."""
        for p, v in self.views.items():
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def zero(self):
        self.flat.zero_()
        self.attach()

    def all_reduce(self):
        """Average gradients over ranks; asynchronous on the side stream (wait() before the optimizer)."""
        w = self.world
        for p, v in self.views.items():     # e.g. optimizer.zero_grad(set_to_none=True) after zero(): autograd allocated a
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():      # fresh .grad outside the flat buffer
                v.copy_(p.grad)
                p.grad = v
        if w == 1 and not (self.always_collective and dist.is_available() and dist.is_initialized()):
            return
        if self.cuda:
            ready = torch.cuda.current_stream().record_event()
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                for lo, hi in self.buckets:
                    chunk = self.flat[lo:hi]
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                    chunk.mul_(1.0 / w)
                self.done = self.side.record_event()
        else:
            for lo, hi in self.buckets:
                chunk = self.flat[lo:hi]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                chunk.mul_(1.0 / w)

    def wait(self):
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None


def shard_indices(n_items, rank, world):
    """Sequence sharding of SURVEY section 8e: rank r gets items r, r+world, ..."""
    return list(range(rank, n_items, world))
