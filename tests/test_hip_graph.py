"""GPU: sequence-level execution (rpg_ramnet_amd/graph.py) — the launch chains of the hot path replayed as hipGraphs must give
the results of the launch-by-launch path (which the other GPU tests pin to the oracle and the reference's goldens).
Reference loops: model/model.py:176-195 (package), trainer/lstm_trainer.py:256-272 (sequence), test.py:212-232 (streaming)."""
import numpy as np
import pytest
import torch

from oracle import ramnet_ref
from recipe import make_item
from util import assert_close, build_hip_model, ref_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,full_frame", [(1, 32, 48, False), (1, 64, 88, False), (2, 36, 50, True)])
def test_latency_stream_equals_eager_primitives(B, H, W, full_frame):
    """graph.LatencyStream (live batch-1 latency: the two fine scales' state updates on a second stream beside the coarse update,
    the residual blocks and decoder 0; four graph replays per measurement) == update_events / update_image / decode launch by launch,
    bit for bit, over an irregular schedule — predictions, final state, after a reset, and with full-frame padding."""
    from rpg_ramnet_amd.graph import LatencyStream
    cfg, _ = ref_cfg("net_seeded_ramnet.npz")
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    model.set_full_frame(full_frame)
    ls = LatencyStream(model, B, H, W)
    rng = np.random.default_rng(4)
    for rnd in range(2):
        st = model.init_states(B, *((H, W) if not full_frame else (model._crop_for(H, W).height_crop_size, model._crop_for(H, W).width_crop_size)))
        for n_ev in [2, 1, 3]:
            item = make_item(rng, B, H, W, n_ev, 5, 1)
            for k in range(n_ev + 1):
                key = "events%d" % k if k < n_ev else "image"
                with torch.no_grad():
                    st, _ = (model.update_events if k < n_ev else model.update_image)(item[key], st)
                    want = model.decode(st, frame_hw=(H, W) if full_frame else None)
                got = (ls.update_events if k < n_ev else ls.update_image)(item[key].to(model.gpu)).clone()
                assert torch.equal(got, want), "LatencyStream differs from the eager launches (%s, round %d)" % (key, rnd)
        for a, b in zip(ls.states, st):
            assert torch.equal(a, b)
        ls.reset()
    model.set_full_frame(False)


@pytest.mark.parametrize("pipelined", [False, True])
def test_graphed_stream_equals_eager_primitives_and_oracle(pipelined):
    """Irregular asynchronous schedule (1..3 event grids between frames), batch 1, persistent state: one hipGraph replay per
    update+decode == update_events / update_image / decode launch by launch (bit-exact), == the oracle to 1e-3."""
    from rpg_ramnet_amd.graph import GraphedStream
    cfg, _ = ref_cfg("net_seeded_ramnet.npz")
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    B, H, W = 1, 32, 48
    gs = GraphedStream(model, B, H, W, pipelined=pipelined)
    rng = np.random.default_rng(2)
    sched = [2, 1, 3, 1]
    st = model.init_states(B, H, W)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ncfg = ramnet_ref.normalize_config(cfg)
    ost = None
    for n_ev in sched:
        item = make_item(rng, B, H, W, n_ev, 5, 1)
        for k in range(n_ev + 1):
            key = "events%d" % k if k < n_ev else "image"
            with torch.no_grad():
                st, _ = (model.update_events if k < n_ev else model.update_image)(item[key], st)
                want = model.decode(st)
            got = gs.wait((gs.update_events if k < n_ev else gs.update_image)(item[key].to(model.gpu))).clone()
            assert torch.equal(got, want), "graph replay differs from the eager launches (%s)" % key
            with torch.no_grad():
                if ost is None:
                    ost = [torch.zeros(B, 64 * 2 ** i, H // 2 ** (i + 1), W // 2 ** (i + 1)) for i in range(3)]
                ost, _ = ramnet_ref._encode(sd, ncfg, "events" if k < n_ev else "images", item[key], ost, None)
                ref = ramnet_ref._decode(sd, ncfg, ost)
            assert_close(got.cpu().numpy(), ref.numpy(), 1e-3, "stream vs oracle " + key)
    for a, b in zip(gs.states, st):
        assert torch.equal(a, b)
    if pipelined:       # back-to-back updates without waiting in between: decodes overlap the next update, results unchanged
        gs.reset()
        st = model.init_states(B, H, W)
        item = make_item(rng, B, H, W, 4, 5, 1)
        preds = []
        for k in range(5):
            key = "events%d" % k if k < 4 else "image"
            with torch.no_grad():
                st, _ = (model.update_events if k < 4 else model.update_image)(item[key], st)
                preds.append(model.decode(st))
        torch.cuda.synchronize()
        dev_item = {k: v.to(model.gpu) for k, v in item.items()}
        got = []
        for k in range(5):
            key = "events%d" % k if k < 4 else "image"
            p = (gs.update_events if k < 4 else gs.update_image)(dev_item[key])
            if k >= 1:                                   # the previous prediction is still intact while this update runs
                got.append(prev.clone())                 # (clone is ordered behind the decode through wait())
            prev = gs.wait(p)
        got.append(prev.clone())
        for a, b in zip(got, preds):
            assert torch.equal(a, b)
    gs.reset()
    assert all(float(s.abs().sum()) == 0.0 for s in gs.states)


def test_graphed_package_equals_model_forward(monkeypatch):
    """Two consecutive packages (state carried): GraphedPackage == model.forward, also with the decoders on a second stream
    (the fork/join is captured into the graph)."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.graph import GraphedPackage
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=3)
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    rng = np.random.default_rng(3)
    items = [make_item(rng, 2, 32, 48, 3, 5, 1) for _ in range(2)]
    for overlap in (False, True):
        ops.set_decoder_overlap(overlap)
        try:
            gp = GraphedPackage(model, items[0])
            prev, lstm = None, ramnet_ref.empty_states_lstm(3)
            for item in items:
                with torch.no_grad():
                    want, supers, lstm = model(item, prev, lstm)
                prev = supers["image"]
                got = gp(item)
                assert list(got.keys()) == list(want.keys())
                for k in want:
                    assert torch.equal(got[k], want[k]), (overlap, k)
        finally:
            ops.set_decoder_overlap(False)


@pytest.mark.parametrize("schedule", ["single_stream", "three_streams"])
def test_graphed_train_step_equals_eager_steps(schedule):
    """Three optimizer steps: graph replays (zero-fill, forward, loss, backward, fold) + eager SGD == the eager path.  The
    second and third step prove that the weight re-packing is INSIDE the graph (the optimizer changed the weights in between)."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.graph import GraphedTrainStep
    from rpg_ramnet_amd.parallel import FlatGradReducer
    from rpg_ramnet_amd.trainer import sequence_loss
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=2, loss_composition=["image", "events1"])
    rng = np.random.default_rng(4)
    seq = [make_item(rng, 2, 32, 48, 2, 5, 1, True, 0.1) for _ in range(2)]
    on = schedule == "three_streams"
    ops.set_wgrad_overlap(on)
    ops.set_decoder_overlap(on)
    try:
        losses, weights = {}, {}
        for mode in ("eager", "graph"):
            model = build_hip_model("ERGB2DepthRecurrent", cfg).train()
            red = FlatGradReducer(model)
            opt = torch.optim.SGD(model.parameters(), lr=0.05)      # linear in the gradient: atomic-order noise stays at its own size
            dseq = [{k: v.to(model.gpu) for k, v in it.items()} for it in seq]
            g = GraphedTrainStep(model, dseq, cfg["loss_composition"], [1, 1], reducer=red) if mode == "graph" else None
            out = []
            for _ in range(3):
                if g is not None:
                    total, _ = g()
                else:
                    red.zero()
                    total, _ = sequence_loss(model, dseq, cfg["loss_composition"], [1, 1])
                    total.backward()
                out.append((float(total), red.flat.clone()))
                opt.step()
            losses[mode] = out
            weights[mode] = torch.cat([p.detach().flatten() for p in model.parameters()])
        names, off = [], 0
        for k, p in reversed(list(model.named_parameters())):          # layout of FlatGradReducer.flat
            names.append((off, off + p.numel(), k))
            off += p.numel()
        for i, ((le, ge), (lg, gg)) in enumerate(zip(losses["eager"], losses["graph"])):
            np.testing.assert_allclose(lg, le, rtol=1e-5, err_msg="loss of step %d" % i)
            scale = float(ge.abs().max())
            worst = sorted(((float((gg[a:b] - ge[a:b]).abs().max()), k) for a, b, k in names), reverse=True)[:3]
            # bulk: stale weight packs in the graph would shift every gradient by ~1e-4 of the scale; isolated entries may differ
            # by one upstream-gradient element (2e-3 of the scale here): after an update whose gradients differ in the order of
            # their atomic sums (1e-10), a ReLU pre-activation within that distance of zero takes the other branch — two EAGER
            # runs differ by exactly such single elements too (measured: decoders.1.conv2d.bias, 3.2e-7 of 2.0e-4)
            assert float((gg - ge).abs().mean()) / scale < 1e-6 * (i + 1), "bulk of the gradients of step %d" % i
            assert worst[0][0] / scale < 5e-3, "gradients of step %d: %s (scale %.2e)" % (i, worst, scale)
        assert float((weights["graph"] - weights["eager"]).abs().max()) < 2e-5
        assert losses["graph"][2][0] != losses["graph"][0][0]          # the steps really trained
    finally:
        ops.set_wgrad_overlap(False)
        ops.set_decoder_overlap(False)


@pytest.mark.parametrize("with_reducer", [False, True])
def test_graphed_train_step_survives_zero_grad_set_to_none(with_reducer):
    """optimizer.zero_grad() (set_to_none=True, the default and what the reference trainer calls) between replays drops
    every .grad; the graph writes through raw pointers captured once.  The runner owns those gradient tensors and re-attaches
    them, so the optimizer keeps training and nothing is written into freed memory."""
    from rpg_ramnet_amd.graph import GraphedTrainStep
    from rpg_ramnet_amd.parallel import FlatGradReducer
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=2, loss_composition=["image", "events1"])
    rng = np.random.default_rng(5)
    seq = [make_item(rng, 1, 32, 48, 2, 5, 1, True, 0.1) for _ in range(2)]
    model = build_hip_model("ERGB2DepthRecurrent", cfg).train()
    red = FlatGradReducer(model) if with_reducer else None
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    dseq = [{k: v.to(model.gpu) for k, v in it.items()} for it in seq]
    g = GraphedTrainStep(model, dseq, cfg["loss_composition"], [1, 1], reducer=red)
    ptrs = {k: p.grad.data_ptr() for k, p in model.named_parameters()}
    losses = []
    for it in range(3):
        opt.zero_grad()                                  # set_to_none=True
        assert all(p.grad is None for p in model.parameters())
        filler = [torch.full((1 << 20,), 7.0, device=model.gpu) for _ in range(8)]      # would land in freed gradient storage
        total, _ = g()
        torch.cuda.synchronize()
        assert np.isfinite(float(total)), "loss of replay %d" % it
        assert all(float(f.min()) == 7.0 and float(f.max()) == 7.0 for f in filler)
        for k, p in model.named_parameters():
            assert p.grad is not None and p.grad.data_ptr() == ptrs[k], k
            assert bool(torch.isfinite(p.grad).all()), "replay %d: %s" % (it, k)
            assert float(p.grad.abs().max()) > 0 or k.endswith("bias"), k
        if red is not None:
            red.all_reduce()
            red.wait()
        opt.step()
        losses.append(float(total))
    assert losses[2] != losses[0]                        # the optimizer stepped on real gradients


def test_branch_streams_equal_serial_updates():
    """ops.set_branch_overlap: the per-scale state updates of one measurement on streams of their own (eager launches, batch 1 and
    2, ConvGRU and ConvLSTM states) == the serial order, bit for bit, over an irregular schedule."""
    from rpg_ramnet_amd import ops
    for tag, B in (("net_seeded_ramnet.npz", 1), ("net_seeded_ramnet_lstm.npz", 2)):
        cfg, _ = ref_cfg(tag)
        model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
        H, W = 64, 96
        rng = np.random.default_rng(8)
        item = make_item(rng, B, H, W, 5, 5, 1)
        outs = {}
        for on in (False, True):
            ops.set_branch_overlap(on)
            try:
                st = model.init_states(B, H, W)
                preds = []
                with torch.no_grad():
                    for key in ("events0", "events1", "image", "events2", "events3", "events4", "image"):
                        st, _ = (model.update_image if key == "image" else model.update_events)(item[key], st)
                        preds.append(model.decode(st))
                torch.cuda.synchronize()
                outs[on] = (preds, st)
            finally:
                ops.set_branch_overlap(False)
        for a, b in zip(outs[False][0], outs[True][0]):
            assert torch.equal(a, b)
        for a, b in zip(outs[False][1], outs[True][1]):
            for u, v in zip(a if isinstance(a, (list, tuple)) else [a], b if isinstance(b, (list, tuple)) else [b]):
                assert torch.equal(u, v)


@pytest.mark.parametrize("tag,B", [("net_seeded_ramnet.npz", 1), ("net_seeded_ramnet_lstm.npz", 2)])
def test_time_batched_stream_equals_eager_primitives(tag, B):
    """graph.TimeBatchedStream: the encoders of the event grids up to the next frame at batch n, the n + 1 decodes as one batched
    chain over a batched state buffer, the state updates one by one per scale — predictions and the carried state are bit-identical
    to update_events / update_image / decode called one by one.  Schedule: a frame with no events before it, groups of 1-3 grids,
    more grids than the group capacity (auto-close without a frame) and a flush at the end of the stream; two passes, so that the
    second one runs on recorded graphs only."""
    from rpg_ramnet_amd.graph import TimeBatchedStream
    cfg, _ = ref_cfg(tag)
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    H, W = 64, 96
    tb = TimeBatchedStream(model, B, H, W, max_events=4)
    rng = np.random.default_rng(16)
    groups = [0, 2, 1, 6, 3, 2]                          # event grids before each frame (6 > max_events: auto-close at 4)
    keys = []
    for n_ev in groups:
        keys += ["events"] * n_ev + ["image"]
    keys += ["events", "events"]                         # trailing grids without a frame: flush()
    data = [torch.from_numpy((rng.standard_normal((B, 5, H, W)) if k == "events" else rng.random((B, 1, H, W))).astype(np.float32)).to(model.gpu)
            for k in keys]
    want, st = [], model.init_states(B, H, W)
    with torch.no_grad():
        for k, x in zip(keys, data):
            st, _ = (model.update_events if k == "events" else model.update_image)(x, st)
            want.append(model.decode(st).clone())
    torch.cuda.synchronize()

    def flat(s):
        return [t for u in s for t in (u if isinstance(u, (list, tuple)) else [u])]

    for run in range(2):
        tb.reset()
        got = []
        for k, x in zip(keys, data):
            out = tb.push_events(x) if k == "events" else tb.push_image(x)
            if out is not None:
                got += list(tb.wait(out).clone())
        out = tb.flush()
        got += list(tb.wait(out).clone())
        assert tb.flush() is None
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), "pass %d: prediction %d (%s)" % (run, i, keys[i])
        for a, b in zip(flat(tb.states), flat(st)):
            assert torch.equal(a, b)


def test_time_batched_stream_follows_weight_updates():
    """The recorded graphs read packed weights by address: after an in-place parameter update (optimizer step, load_state_dict) the
    runtime records them again instead of replaying stale — or freed — packs."""
    from rpg_ramnet_amd.graph import TimeBatchedStream
    cfg, _ = ref_cfg("net_seeded_ramnet.npz")
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval()
    B, H, W = 1, 32, 48
    tb = TimeBatchedStream(model, B, H, W, max_events=3)
    rng = np.random.default_rng(26)
    ev = [torch.from_numpy(rng.standard_normal((B, 5, H, W)).astype(np.float32)).to(model.gpu) for _ in range(2)]
    im = torch.from_numpy(rng.random((B, 1, H, W)).astype(np.float32)).to(model.gpu)

    def eager():
        st = model.init_states(B, H, W)
        out = []
        with torch.no_grad():
            for x in ev:
                st, _ = model.update_events(x, st)
                out.append(model.decode(st).clone())
            st, _ = model.update_image(im, st)
            out.append(model.decode(st).clone())
        return out

    def batched():
        tb.reset()
        for x in ev:
            tb.push_events(x)
        return list(tb.wait(tb.push_image(im)).clone())

    for a, b in zip(batched(), eager()):
        assert torch.equal(a, b)
    before = batched()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.05)
    after = batched()
    assert not torch.equal(after[-1], before[-1])
    for a, b in zip(after, eager()):
        assert torch.equal(a, b)


def test_time_batched_stream_full_frame_equals_eager_primitives():
    """Full-frame mode through the streaming runtime (what inference.stream_dataset runs on raw 260 x 346 recordings): a 26 x 35 stream
    with model.set_full_frame() — inputs reflect-padded in the repack, states at 32 x 40, predictions cropped — gives bit-identical
    predictions to update_events / update_image / decode(frame_hw=) called one by one."""
    from rpg_ramnet_amd.graph import TimeBatchedStream
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=2)
    model = build_hip_model("ERGB2DepthRecurrent", cfg).eval().set_full_frame(True)
    rng = np.random.default_rng(5)
    Hh, W = 26, 35
    groups = [[torch.from_numpy(rng.standard_normal((1, 5, Hh, W)).astype(np.float32)) for _ in range(n)] for n in (2, 1, 3)]
    frames = [torch.from_numpy(rng.random((1, 1, Hh, W)).astype(np.float32)) for _ in groups]
    tb = TimeBatchedStream(model, 1, Hh, W, max_events=4)
    st = model.init_states(1, 32, 40)
    with torch.no_grad():
        for evs, fr in zip(groups, frames):
            want = []
            for e in evs:
                st, _ = model.update_events(e, st)
                want.append(model.decode(st, frame_hw=(Hh, W)).clone())
            st, _ = model.update_image(fr, st)
            want.append(model.decode(st, frame_hw=(Hh, W)).clone())
            for e in evs:
                tb.push_events(e)
            got = tb.wait(tb.push_image(fr))
            torch.cuda.synchronize()
            assert got.shape == (len(evs) + 1, 1, 1, Hh, W)
            for a, b in zip(got, want):
                assert torch.equal(a, b)
