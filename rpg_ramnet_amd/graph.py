"""Sequence-level execution: the launch chains of the RAM-Net hot path captured as hipGraphs.

The reference drives three nested Python loops — sequence (trainer/lstm_trainer.py:256-272), package (model/model.py:176-195),
layer — and so does the eager path here: ~150 kernel launches per data package, each a ctypes call.  That is invisible at B=8
(the GPU is the bottleneck) and dominant at batch 1 (streaming inference, BASELINE configs[3]: ~20 launches of a few
microseconds each per update).  This module records those chains ONCE with HIP stream capture (torch.cuda.CUDAGraph: the
library launches on torch's current stream, so its kernels, memsets and the side-stream forks are captured as they are) and
replays them with one host call:

* ``GraphedStream``    — asynchronous inference with a persistent multi-scale state: one graph per (modality update + decode),
                         state kept in static device buffers (test.py:212-232 call pattern; configs[3]).
* ``GraphedPackage``   — one data package, K event updates + 1 frame update + K+1 decodes (model.py:176-219), state carried.
* ``GraphedTrainStep`` — gradient zero-fill, forward over the L packages of a sequence, loss assembly, BPTT backward and the
                         weight-gradient fold of one optimizer step (lstm_trainer.py:228-390); the optimizer and the gradient
                         all-reduce stay outside (they bump parameter versions / use the collective library).

Replays are bit-identical to the eager path for forward results; gradients differ only by the order of the atomic partial sums
(as between two eager runs).  Static input buffers are filled with ``copy_`` (device-to-device or pinned host-to-device).
"""
import torch

from . import ops
from .trainer import empty_states_lstm, sequence_loss


def _warm(fn, n=2):
    """Eager runs before capture: lazy allocations (packs, gradient workspaces, rocBLAS handles, LDS attributes) happen here."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


class GraphedStream:
    """Streaming inference (batch B, persistent state): ``update_events(grid)`` / ``update_image(frame)`` fold one measurement
    into the shared state and return the depth prediction decoded from it — hipGraph replays, no per-kernel host work.

    The state lives in two static buffer sets used in ping-pong (update k reads set s and writes set 1-s: a previous state is
    never modified in place, as in the eager path).  ``pipelined=True`` replays the decoder graph of update k on a second HIP
    stream, concurrent with update k+1 on the first: at batch 1 both chains are latency-bound with a fraction of the chip
    occupied each (a dozen workgroups per launch), so they overlap almost freely.  Predictions are then valid on the caller's
    stream after ``wait(pred)`` (an event wait, no host sync) and until the second-next update overwrites the buffer."""

    def __init__(self, model, B, H, W, pipelined=False, branches=True):
        """branches: capture the state updates of the three scales as parallel branches of the update graph (they are mutually
        independent, ops.set_branch_overlap) — the graph's critical path is then head + encoders + ONE state update."""
        assert not bool(model.baseline), "streaming graphs are built for the asynchronous RAM-Net (not the baselines)"
        self.model, dev = model, model.gpu
        self.pipelined = pipelined
        branch0 = ops.branch_overlap()
        ops.set_branch_overlap(bool(branches))
        self.ev_in = torch.zeros(B, model.num_bins_events, H, W, device=dev)
        self.im_in = torch.zeros(B, model.num_bins_rgb, H, W, device=dev)
        self.sets = [model.init_states(B, H, W), model.init_states(B, H, W)]
        self.cur = 0                                    # index of the set holding the current state
        self.upd, self.dec, self.pred = {}, [None, None], [None, None]
        self.side = torch.cuda.Stream(device=dev) if pipelined else None
        self.dec_done = [None, None]                    # event behind the last decode that READ set i
        was_training = model.training
        model.eval()
        for src in (0, 1):
            for kind, buf, update in (("events", self.ev_in, model.update_events), ("image", self.im_in, model.update_image)):
                def run(buf=buf, update=update, src=src):
                    with torch.no_grad():
                        new, _ = update(buf, self.sets[src])
                        for dst, s_ in zip(self.sets[1 - src], new):
                            for d, t in zip(self._flat(dst), self._flat(s_)):
                                d.copy_(t)
                _warm(run)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                self.upd[(kind, src)] = g

            def decode(src=src):
                with torch.no_grad():
                    return model.decode(self.sets[src])
            _warm(decode)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.pred[src] = decode()
            self.dec[src] = g
        self.reset()
        model.train(was_training)
        ops.set_branch_overlap(branch0)

    @staticmethod
    def _flat(s):
        return list(s) if isinstance(s, (list, tuple)) else [s]

    @property
    def states(self):
        """The current multi-scale state (static buffers; valid on the caller's stream)."""
        return self.sets[self.cur]

    def reset(self):
        torch.cuda.synchronize(self.model.gpu)
        for st in self.sets:
            for s in st:
                for t in self._flat(s):
                    t.zero_()
        self.cur, self.dec_done = 0, [None, None]

    def _step(self, kind, buf, data):
        src, dst = self.cur, 1 - self.cur
        main = torch.cuda.current_stream()
        if self.dec_done[dst] is not None:              # the decode that still reads the set this update overwrites
            main.wait_event(self.dec_done[dst])
        buf.copy_(data, non_blocking=True)
        self.upd[(kind, src)].replay()
        self.cur = dst
        if not self.pipelined:
            self.dec[dst].replay()
            return self.pred[dst]
        self.side.wait_event(main.record_event())
        with torch.cuda.stream(self.side):
            self.dec[dst].replay()
            self.dec_done[dst] = self.side.record_event()
        return self.pred[dst]

    def update_events(self, grid):
        """grid [B, Ce, H, W] (device or pinned host) -> prediction [B,1,H,W] (a static buffer; pipelined: see wait())."""
        return self._step("events", self.ev_in, grid)

    def update_image(self, frame):
        return self._step("image", self.im_in, frame)

    def wait(self, pred=None):
        """Make the caller's stream wait for the decodes in flight (pipelined mode); returns `pred`."""
        for e in self.dec_done:
            if e is not None:
                torch.cuda.current_stream().wait_event(e)
        return pred


class GraphedPackage:
    """One data package through ``model.forward`` (K event grids + 1 frame -> K+1 predictions), state carried in static
    buffers across calls; ``reset()`` starts a new recording."""

    def __init__(self, model, example_item):
        self.model = model
        K = model.every_x_rgb_frame
        self.item = {k: torch.empty_like(v, device=model.gpu).copy_(v) for k, v in example_item.items() if not k.startswith("depth_")}
        B, _, H, W = self.item["image"].shape
        self.states = model.init_states(B, H, W)
        was_training = model.training
        model.eval()

        def run():
            with torch.no_grad():
                preds, supers, _ = model(self.item, [s.permute(0, 3, 1, 2) for s in self.states], empty_states_lstm(K))
                for dst, src in zip(self.states, supers["image"]):
                    dst.copy_(src.permute(0, 2, 3, 1))
            return preds
        assert model.state_combination == "convgru" and model.recurrent_block_type == "conv", "graphed package: ConvGRU RAM-Net"
        _warm(run)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.preds = run()
        self.reset()
        model.train(was_training)

    def reset(self):
        for s in self.states:
            s.zero_()

    def __call__(self, item):
        for k, buf in self.item.items():
            buf.copy_(item[k], non_blocking=True)
        self.graph.replay()
        return self.preds


class GraphedTrainStep:
    """zero gradients -> forward over the sequence -> loss (trainer.sequence_loss) -> BPTT backward -> weight-gradient fold, as
    one hipGraph.  ``sequence`` provides the static input buffers (refill them in place, e.g. ``step.load(new_sequence)``);
    the gradients land in the parameters' ``.grad`` (the flat buffer of ``parallel.FlatGradReducer`` when one is attached).  The
    gradient tensors of the capture are owned by this object and re-installed as ``.grad`` at every call, so an
    ``optimizer.zero_grad()`` (set_to_none=True) between replays is harmless; do not REPLACE ``.grad`` by other tensors and expect
    the replay to fill those."""

    def __init__(self, model, sequence, loss_composition, loss_weights, reducer=None, grad_loss_weight=None, warmup=2):
        self.model, self.reducer = model, reducer
        self.sequence = [{k: v.to(model.gpu) for k, v in item.items()} for item in sequence]
        assert model.training, "call model.train() first"

        def zero():
            if reducer is not None:
                reducer.zero()
            else:
                for p in model.parameters():
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    else:
                        p.grad.zero_()

        def run():
            zero()
            total, reported = sequence_loss(model, self.sequence, loss_composition, loss_weights, grad_loss_weight=grad_loss_weight)
            total.backward()
            return total.detach(), reported

        _warm(run, warmup)
        torch.cuda.empty_cache()
        ops.invalidate_packs()            # every weight pack is re-run — and therefore recorded — inside the capture
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.total, self.reported = run()
        ops.invalidate_packs()            # eager code must not trust packs whose refresh now lives in the graph
        # the graph holds RAW POINTERS to the gradient tensors of the capture: keep them alive, and re-attach them as .grad when the
        # caller dropped them (optimizer.zero_grad() defaults to set_to_none=True, as the reference trainer's call does) — a replay
        # into storage the caching allocator took back would corrupt other tensors and leave the optimizer with nothing to step on
        self.grads = [(p, p.grad) for p in model.parameters() if p.grad is not None]

    def _attach(self):
        if self.reducer is not None:
            self.reducer.attach()
        for p, g in self.grads:
            if p.grad is not g:
                p.grad = g

    def load(self, sequence):
        for dst, src in zip(self.sequence, sequence):
            for k, buf in dst.items():
                buf.copy_(src[k], non_blocking=True)

    def __call__(self):
        """Replay; returns (differentiated loss, reported loss) as device scalars (static buffers)."""
        self._attach()
        self.graph.replay()
        return self.total, self.reported
