"""Sequence-level execution: the launch chains of the RAM-Net hot path captured as hipGraphs.

The reference drives three nested Python loops — sequence (trainer/lstm_trainer.py:256-272), package (model/model.py:176-195),
layer — and so does the eager path here: ~150 kernel launches per data package, each a ctypes call.  That is invisible at B=8
(the GPU is the bottleneck) and dominant at batch 1 (streaming inference, BASELINE configs[3]: ~20 launches of a few
microseconds each per update).  This module records those chains ONCE with HIP stream capture (torch.cuda.CUDAGraph: the
library launches on torch's current stream, so its kernels, memsets and the side-stream forks are captured as they are) and
replays them with one host call:

* ``GraphedStream``    — asynchronous inference with a persistent multi-scale state: one graph per (modality update + decode),
                         state kept in static device buffers (test.py:212-232 call pattern; configs[3]).
* ``TimeBatchedStream`` — a recorded stream in groups of consecutive measurements: encoders and decoders of a group at batch n / n+1
                         (they do not depend on the order of the updates), the state updates one by one per scale.
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


class LatencyStream:
    """Live batch-1 streaming with the shortest path from a measurement to its depth map (BASELINE configs[3], one measurement in, one
    prediction out, the caller waits for it): the state updates of the two FINE scales are not on that path — the decoder needs them
    only for its last two skip connections — so they run on a second HIP stream beside the coarse scale's update, the residual blocks
    and the first decoder.  Four hipGraph replays per measurement on two real streams (no forks inside a capture):

        main:  A  = head, encoder 0, encoder 1                 C1 = encoder 2, update of scale 2, residual blocks, decoder 0
        side:        B  = updates of scales 0 and 1            C2 = decoders 1, 2 + prediction   (main, after B)

    Same kernels, same arithmetic as ``GraphedStream`` (bit-identical predictions and states); state in two buffer sets used in
    ping-pong.  RAM-Net wiring only (shared state, plain convolutional encoders: statenet.py:215-237)."""

    def __init__(self, model, B, H, W):
        net = model.statenetphasedrecurrent
        assert not bool(model.baseline) and net.recurrent_block_type == 'conv' and net.num_encoders == 3
        self.model, dev = model, model.gpu
        self.ev_in = torch.zeros(B, model.num_bins_events, H, W, device=dev)
        self.im_in = torch.zeros(B, model.num_bins_rgb, H, W, device=dev)
        crop = model._crop_for(H, W)                 # full-frame mode: states at the reflect-padded size, predictions cropped back
        Hs, Ws = (crop.height_crop_size, crop.width_crop_size) if crop is not None else (H, W)
        self.sets = [model.init_states(B, Hs, Ws), model.init_states(B, Hs, Ws)]
        self.cur = 0
        self.side = torch.cuda.Stream(device=dev)
        # (the critical chain on a high-priority stream of its own was measured: 0.54 -> 0.93 ms, profiles/r04_h_tuning_notes.md)
        self.gA, self.gB, self.gC1, self.gC2, self.pred = {}, {}, {}, {}, {}
        pair = net.state_combination == 'convlstm'
        pick = (lambda s: s[0]) if pair else (lambda s: s)
        was_training = model.training
        model.eval()
        feats = {}

        def capture(fn):
            _warm(fn)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            return g, out
        for kind, buf, head, encs, combs in (("events", self.ev_in, net.head_events, net.encoders_events, net.state_combination_events),
                                             ("image", self.im_in, net.head_rgb, net.encoders_rgb, net.state_combination_images)):
            def part_a(buf=buf, head=head, encs=encs):
                with torch.no_grad():
                    x0 = encs[0](head(ops.pack_input(buf, dev, model._crop_for(H, W))))
                    return x0, encs[1](x0)
            self.gA[kind], feats[kind] = capture(part_a)
            for src in (0, 1):
                def part_b(kind=kind, combs=combs, src=src):
                    with torch.no_grad():
                        for i in (0, 1):
                            combs[i](feats[kind][i], self.sets[src][i], self.sets[1 - src][i])
                self.gB[(kind, src)], _ = capture(part_b)

                def part_c1(kind=kind, encs=encs, combs=combs, src=src):
                    with torch.no_grad():
                        combs[2](encs[2](feats[kind][1]), self.sets[src][2], self.sets[1 - src][2])
                        x = pick(self.sets[1 - src][2])
                        for rb in net.resblocks:
                            x = rb(x)
                        return net.decoders[0](x)
                self.gC1[(kind, src)], d0 = capture(part_c1)

                def part_c2(d0=d0, dst=1 - src):     # (one per (modality, parity): it reads decoder 0's output out of THAT C1 graph's pool)
                    with torch.no_grad():
                        x = net.decoders[1](d0, pick(self.sets[dst][1]))
                        x = net.decoders[2](x, pick(self.sets[dst][0]))
                        if net.norm in ('BN', 'IN'):
                            p = net.pred(x, act='sigmoid').permute(0, 3, 1, 2)
                        else:
                            p = ops.PredSigmoid.apply(x, net.pred.conv2d.weight, net.pred.conv2d.bias)
                        crop = model._crop_for(H, W)
                        return p if crop is None else crop.crop(p)
                self.gC2[(kind, src)], self.pred[(kind, src)] = capture(part_c2)
        self._pick_side_stream()
        self.reset()
        model.train(was_training)

    def _pick_side_stream(self, candidates=4, reps=8):
        """HIP maps streams onto a few hardware queues; a side stream that lands on the caller's queue runs the fine updates IN ORDER
        with the critical chain (measured: 0.62 instead of 0.54 ms).  Time a few measurements with each of `candidates` streams, keep the best."""
        import time
        best = None
        for i in range(candidates):
            cand = self.side if i == 0 else torch.cuda.Stream(device=self.model.gpu)
            self.side = cand
            for r in range(reps + 2):
                if r == 2:
                    torch.cuda.synchronize(self.model.gpu)
                    t = time.perf_counter()
                self._step("events", self.ev_in, self.ev_in)
                torch.cuda.synchronize(self.model.gpu)
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, cand)
        self.side = best[1]

    @property
    def states(self):
        return self.sets[self.cur]

    def reset(self):
        torch.cuda.synchronize(self.model.gpu)
        for st in self.sets:
            for s in st:
                for t in (list(s) if isinstance(s, (list, tuple)) else [s]):
                    t.zero_()
        self.cur = 0

    def _step(self, kind, buf, data):
        src, dst = self.cur, 1 - self.cur
        main = torch.cuda.current_stream()
        buf.copy_(data, non_blocking=True)
        self.gA[kind].replay()
        self.side.wait_event(main.record_event())
        with torch.cuda.stream(self.side):
            self.gB[(kind, src)].replay()
            done_b = self.side.record_event()
        self.gC1[(kind, src)].replay()
        main.wait_event(done_b)
        self.gC2[(kind, src)].replay()
        self.cur = dst
        return self.pred[(kind, src)]

    def update_events(self, grid):
        """grid [B, Ce, H, W] (device or pinned host) -> prediction [B,1,H,W] (a static buffer, valid on the caller's stream until the
        second-next measurement)."""
        return self._step("events", self.ev_in, grid)

    def update_image(self, frame):
        return self._step("image", self.im_in, frame)


class GraphedStream:
    """Streaming inference (batch B, persistent state): ``update_events(grid)`` / ``update_image(frame)`` fold one measurement
    into the shared state and return the depth prediction decoded from it — hipGraph replays, no per-kernel host work.

    The state lives in two static buffer sets used in ping-pong (update k reads set s and writes set 1-s: a previous state is
    never modified in place, as in the eager path).  ``pipelined=True`` replays the decoder graph of update k on a second HIP
    stream, concurrent with update k+1 on the first: at batch 1 both chains are latency-bound with a fraction of the chip
    occupied each (a dozen workgroups per launch), so they overlap almost freely.  Predictions are then valid on the caller's
    stream after ``wait(pred)`` (an event wait, no host sync) and until the second-next update overwrites the buffer."""

    def __init__(self, model, B, H, W, pipelined=False):
        assert not bool(model.baseline), "streaming graphs are built for the asynchronous RAM-Net (not the baselines)"
        self.model, dev = model, model.gpu
        self.pipelined = pipelined
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
                    with torch.no_grad():                         # the cells write the other set themselves: no copy kernels in the chain
                        update(buf, self.sets[src], out=self.sets[1 - src])
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


class TimeBatchedStream:
    """Throughput runtime for a recorded asynchronous stream that exploits what RAM-Net's wiring allows and the reference's loop
    (test.py:205-232, model.py:176-195) does not: only the state updates are sequential in time.

    * The head and the strided encoders of a measurement do not see the state (statenet.py:215-237: the encoder feature x chains
      through the scales, the shared state is only updated), so the encoders of a GROUP of consecutive measurements — the n event
      grids up to the next frame — run as ONE launch chain at batch n (the frame at batch 1).
    * The decoder of measurement k reads the state after update k and nothing else, so the n + 1 decodes of the group run as ONE
      chain at batch n + 1 over a batched state buffer [n + 1, h, w, C] per scale that the updates fill slot by slot.
    * In between, the state updates run one by one, each scale as an independent chain on a stream of its own (the recurrence of a
      scale involves that scale only); every update writes its new state straight into its slot (GRUCell / LSTMCell `out`).
    At batch 1 a launch of this path occupies 44-350 workgroups of a 256-CU chip and is bound by the latency of its own chain;
    batching over time gives the encoder and decoder launches the grid sizes (and the efficiency) of the training shapes and
    takes 2/3 of the launches off the critical path.  Groups are pipelined: encoders of group g+1 | updates of group g | decodes
    of group g-1 (hipGraph replays — one graph per chain, each on a HIP stream of its own: encoders, one update chain per scale,
    two alternating decode streams —, two buffer parities, event waits, no host sync).
    Results are bit-identical to update_events / update_image / decode called one by one (every kernel computes a batch element
    independently of the others).  The graphs are recorded lazily per group shape and recorded again when a parameter of the model
    has been modified in place since (optimizer step, load_state_dict): they read the packed weights by address.  ``push_events`` buffers a grid, ``push_image`` buffers the frame and closes the group,
    ``flush`` closes a group without a frame; both return the group's predictions [n (+1), B, 1, H, W] as a static buffer that
    stays valid (after ``wait``) until the second-next group of the same shape."""

    def __init__(self, model, B, H, W, max_events=8):
        assert not bool(model.baseline) and model.recurrent_block_type == "conv", "time-batched stream: asynchronous RAM-Net, conv encoders"
        if getattr(model, "norm", None) in ("BN", "IN") and model.training:
            raise RuntimeError("time-batched stream with norm layers: eval mode only (batching the encoders / decoders over time would "
                               "change the batch statistics of a training-mode BatchNorm)")
        self.model, self.B, self.H, self.W, self.NM, dev = model, B, H, W, max_events, model.gpu
        self.net = net = model.statenetphasedrecurrent
        T = max_events + 1
        self.in_ev = [torch.zeros(max_events * B, model.num_bins_events, H, W, device=dev) for _ in range(2)]
        self.in_im = [torch.zeros(B, model.num_bins_rgb, H, W, device=dev) for _ in range(2)]
        # full-frame mode (model.set_full_frame(): raw 260 x 346 frames): the inputs are reflect-padded inside the repack, the states and
        # the decoders run at the padded size, the returned predictions are the cropped window of the static buffer
        self.crop = model._crop_for(H, W)
        Hs, Ws = (H, W) if self.crop is None else (self.crop.height_crop_size, self.crop.width_crop_size)
        self.carry = model.init_states(B, Hs, Ws)
        self.pair = isinstance(self.carry[0], (list, tuple))

        def batched(s_):
            return [torch.zeros((T * B,) + tuple(t.shape[1:]), device=dev) for t in s_] if self.pair else torch.zeros((T * B,) + tuple(s_.shape[1:]), device=dev)
        self.S = [[batched(s_) for s_ in self.carry] for _ in range(2)]
        self.SE = torch.cuda.Stream(device=dev)
        self.SG = [torch.cuda.Stream(device=dev) for _ in range(net.num_encoders)]       # one update chain per scale
        self.SD = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        self.graphs, self.feat, self.pred = {}, {}, {}
        self._wv = self._weights_version()
        self.reset()

    # ---- the three stages of a group (n event grids, f = 1 if a frame closes it) on buffer parity p
    def _enc(self, n, f, p):
        net, B, dev = self.net, self.B, self.model.gpu
        fe = fi = None
        with torch.no_grad():
            if n:
                x = net.head_events(ops.pack_input(self.in_ev[p][:n * B], dev, self.crop))
                fe = []
                for e in net.encoders_events:
                    x = e(x)
                    fe.append(x)
            if f:
                x = net.head_rgb(ops.pack_input(self.in_im[p], dev, self.crop))
                fi = []
                for e in net.encoders_rgb:
                    x = e(x)
                    fi.append(x)
        return fe, fi

    def _slot(self, s_, j):
        B = self.B
        return [t[j * B:(j + 1) * B] for t in s_] if self.pair else s_[j * B:(j + 1) * B]

    def _upd(self, n, f, p, i):
        """The n + f state updates of scale i, one by one (the recurrence of a scale involves that scale only): each writes its
        new state into slot j of the batched state buffer, the last one is copied into the carried state."""
        net, B = self.net, self.B
        fe, fi = self.feat[(n, f, p)]
        with torch.no_grad():
            h = self.carry[i]
            for j in range(n + f):
                dst = self._slot(self.S[p][i], j)
                comb = net.state_combination_events[i] if j < n else net.state_combination_images[i]
                comb(fe[i][j * B:(j + 1) * B] if j < n else fi[i], h, dst)
                h = dst
            for d, t in zip(self.carry[i] if self.pair else [self.carry[i]], h if self.pair else [h]):
                d.copy_(t)

    def _dec(self, n, f, p):
        m = (n + f) * self.B
        with torch.no_grad():
            return self.model.decode([[t[:m] for t in s_] if self.pair else s_[:m] for s_ in self.S[p]])

    def _capture(self, key):
        """First group of this shape on this parity: record its three graphs.  The warm-up runs execute the stages for real, so the
        carried state is saved and restored around them."""
        n, f, p = key
        torch.cuda.synchronize(self.model.gpu)
        was_training = self.model.training
        self.model.eval()
        flat = [t for s_ in self.carry for t in (s_ if self.pair else [s_])]
        saved = [t.clone() for t in flat]
        stages = [(lambda: self._enc(n, f, p), self.feat)]
        stages += [(lambda i=i: self._upd(n, f, p, i), None) for i in range(self.net.num_encoders)]      # one graph per scale
        stages += [(lambda: self._dec(n, f, p), self.pred)]
        gs = []
        for stage, store in stages:
            _warm(stage)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = stage()
            if store is not None:
                store[key] = out
            gs.append(g)
        for t, v in zip(flat, saved):
            t.copy_(v)
        torch.cuda.synchronize(self.model.gpu)
        self.model.train(was_training)
        self.graphs[key] = gs

    def reset(self):
        torch.cuda.synchronize(self.model.gpu)
        for s_ in self.carry:
            for t in (s_ if self.pair else [s_]):
                t.zero_()
        self.p, self.n, self.opened = 0, 0, False
        self.g_done, self.d_done = [None, None], [None, None]

    @property
    def states(self):
        """The current multi-scale state (valid after wait())."""
        return self.carry

    def _open(self):
        """First measurement of a group: its input and feature buffers are free once the updates of the group two back are done."""
        if not self.opened:
            self.SE.wait_stream(torch.cuda.current_stream())
            for e in self.g_done[self.p] or ():
                self.SE.wait_event(e)
            self.opened = True

    def push_events(self, grid):
        """Buffer one event voxel grid [B, Ce, H, W]; returns the group's predictions when this grid fills the group, else None."""
        self._open()
        B = self.B
        self.SE.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.SE):
            self.in_ev[self.p][self.n * B:(self.n + 1) * B].copy_(grid, non_blocking=True)
        if torch.is_tensor(grid) and grid.is_cuda:
            grid.record_stream(self.SE)
        self.n += 1
        return self._close(0) if self.n == self.NM else None

    def push_image(self, frame):
        """Buffer the frame [B, Cr, H, W] that closes the group; returns the predictions [n + 1, B, 1, H, W] of the group."""
        self._open()
        self.SE.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.SE):
            self.in_im[self.p].copy_(frame, non_blocking=True)
        if torch.is_tensor(frame) and frame.is_cuda:
            frame.record_stream(self.SE)
        return self._close(1)

    def flush(self):
        """Close a group that has no frame (end of the stream); None if nothing is buffered."""
        return self._close(0) if self.n else None

    def _weights_version(self):
        return sum(p._version for p in self.model.parameters())

    def _close(self, f):
        key = (self.n, f, self.p)
        n, p = self.n, self.p
        v = self._weights_version()
        if v != self._wv:           # the recorded graphs read weight packs made for the old values (and freed since): record again
            torch.cuda.synchronize(self.model.gpu)
            self.graphs, self.feat, self.pred, self._wv = {}, {}, {}, v
        if key not in self.graphs:
            self._capture(key)
            self.opened = False
            self._open()
        gs = self.graphs[key]
        ge, gd = gs[0], gs[-1]
        with torch.cuda.stream(self.SE):
            ge.replay()
            e_done = self.SE.record_event()
        done = []
        for SG, gg in zip(self.SG, gs[1:-1]):
            SG.wait_event(e_done)
            if self.d_done[p] is not None:               # the decode that still reads this parity's batched states
                SG.wait_event(self.d_done[p])
            with torch.cuda.stream(SG):
                gg.replay()
                done.append(SG.record_event())
        self.g_done[p] = done
        D = self.SD[p]
        for e in done:
            D.wait_event(e)
        with torch.cuda.stream(D):
            gd.replay()
            self.d_done[p] = D.record_event()
        self.p, self.n, self.opened = 1 - p, 0, False
        if self.crop is not None:
            c = self.crop
            return c.crop(self.pred[key].view(n + f, self.B, 1, c.height_crop_size, c.width_crop_size))
        return self.pred[key].view(n + f, self.B, 1, self.H, self.W)

    def wait(self, pred=None):
        """Make the caller's stream wait for everything in flight; returns `pred`."""
        cur = torch.cuda.current_stream()
        for e in [x for g in self.g_done for x in (g or ())] + self.d_done:
            if e is not None:
                cur.wait_event(e)
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
                reducer.zero(arm=False)          # (the collectives stay outside the graph: all_reduce() after the replay)
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
