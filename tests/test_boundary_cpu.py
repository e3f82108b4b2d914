"""CPU (no GPU): the drop-in boundary — C ABI exports, module API, config behaviour, state_dict layout, host logic."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from util import load_golden, ref_cfg
import torch_restatements as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rpg_ramnet_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "ramnet_hip.h")).read()
    declared = set(re.findall(r"\b(ramnet_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(_hip.EXPORTS), "ctypes table and header disagree: %s" % (declared ^ set(_hip.EXPORTS))
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ramnet_abi_version.restype = ctypes.c_int
    assert lib.ramnet_abi_version() == int(re.search(r"#define RAMNET_ABI_VERSION (\d+)", hdr).group(1))


def test_desc_struct_layout_matches_header_field_order():
    from rpg_ramnet_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "ramnet_hip.h")).read()
    for cname, cls in (("ramnet_conv_desc", _hip.ConvDesc), ("ramnet_wgrad_desc", _hip.WgradDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(float|int|int8_t|uint8_t|size_t|ramnet_wgrad_seg)\s+", "", decl)
            names += [re.sub(r"[\*\s]|\[\d+\]", "", n) for n in decl.split(",")]
        assert names == [f[0] for f in cls._fields_], cname


def test_missing_library_fails_loudly(monkeypatch):
    from rpg_ramnet_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/librpg_ramnet_hip.so")
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.lib()


@pytest.mark.parametrize("tag,arch", [("seeded_ramnet", "ERGB2DepthRecurrent"), ("seeded_ramnet_lstm", "ERGB2DepthRecurrent"),
                                      ("seeded_base_rgb", "ERGB2DepthRecurrent"), ("seeded_unet", "ERGB2Depth")])
def test_state_dict_layout_and_seeded_init_match_reference(tag, arch):
    """Keys/shapes equal the reference's checkpoint layout and torch.manual_seed(0) gives the reference's weights."""
    from rpg_ramnet_amd.model import model as mm
    cfg, z = ref_cfg("net_%s.npz" % tag)
    torch.manual_seed(0)
    m = getattr(mm, arch)(cfg)
    ref_keys = [k[5:] for k in z.files if k.startswith("wsum.")]
    assert list(m.state_dict().keys()) == ref_keys
    for k, v in m.state_dict().items():
        got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
        np.testing.assert_allclose(got, z["wsum." + k], rtol=1e-6, atol=1e-9, err_msg=k)
    assert isinstance(m, torch.nn.Module) and hasattr(m, "config") and hasattr(m, "logger")
    m.summary()


def test_explicit_weight_fixture_loads_strict():
    """A reference state_dict (narrow model) loads with strict=True."""
    from rpg_ramnet_amd.model import model as mm
    z = load_golden("net_small_gru.npz")
    cfg = json.loads(str(z["config"]))
    m = mm.ERGB2DepthRecurrent(cfg)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    m.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("tag,arch", [("small_gru_bn", "ERGB2DepthRecurrent"), ("small_gru_in", "ERGB2DepthRecurrent"),
                                      ("small_lstm_tconv_bn", "ERGB2DepthRecurrent"), ("small_unet_bn", "ERGB2Depth"),
                                      ("small_unet_in", "ERGB2Depth")])
def test_norm_variants_state_dict_and_seeded_init(tag, arch):
    """`norm: "BN" | "IN"`: same state_dict keys / shapes / dtypes as the reference (norm_layer.* / bn1.* / bn2.* incl. the running
    buffers and num_batches_tracked; BatchNorm convolutions without bias) and bit-identical torch.manual_seed(0) initialisation."""
    from rpg_ramnet_amd.model import model as mm
    z = load_golden("norm_%s.npz" % tag)
    cfg = json.loads(str(z["config"]))
    torch.manual_seed(0)
    m = getattr(mm, arch)(cfg)
    ref = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert v.dtype == ref[k].dtype and v.shape == ref[k].shape, k
        assert torch.equal(v, ref[k]), k
    m.load_state_dict(ref, strict=True)
    if cfg["norm"] == "BN":
        assert not any(k.endswith("conv2d.bias") for k in sd if ".decoders." in k or ".resblocks." in k or ".pred." in k)


def test_constructor_contract_errors():
    from rpg_ramnet_amd.model import model as mm
    base = dict(num_bins_rgb=1, num_bins_events=5, gpu=0, state_combination="convgru", num_encoders=3)
    with pytest.raises(AssertionError):
        mm.ERGB2DepthRecurrent({"num_bins_events": 5, "gpu": 0})
    with pytest.raises(KeyError):
        mm.ERGB2DepthRecurrent({k: v for k, v in base.items() if k != "gpu"})        # model.py:77
    with pytest.raises(KeyError):
        mm.ERGB2DepthRecurrent(dict(base, skip_type="bogus"))                         # statenet.py:57-59
    with pytest.raises(KeyError):
        mm.ERGB2DepthRecurrent(dict(base, state_combination="bogus"))                 # statenet.py:69-71
    assert mm.ERGB2DepthRecurrent(dict(base, norm="none")).statenetphasedrecurrent.pred.conv2d.bias is not None      # any other string = no norm
    m = mm.ERGB2DepthRecurrent(dict(base, spatial_resolution=[112, 112]))              # extra keys ignored
    assert m.num_residual_blocks == 2 and m.base_num_channels == 32 and m.every_x_rgb_frame == 1
    assert m.recurrent_block_type == "convlstm" and m.use_upsample_conv is True       # reference defaults


def test_tap_lists_reproduce_conv_and_transposed_conv():
    """Host logic: the tap lists fed to the kernels, evaluated in numpy, equal F.conv2d / its input gradient."""
    from rpg_ramnet_amd.ops import Taps
    rng = np.random.default_rng(0)

    def taps(t):
        return [(t.dy[i], t.dx[i], t.wt[i]) for i in range(t.n)]

    for k, stride in [(5, 1), (3, 1), (5, 2)]:
        pad = k // 2
        H, W = 9, 11
        x = rng.standard_normal((H, W))
        w = rng.standard_normal((k, k))
        xt = torch.from_numpy(x)[None, None].requires_grad_(True)
        y = torch.nn.functional.conv2d(xt, torch.from_numpy(w)[None, None], None, stride, pad)
        Ho, Wo = y.shape[2:]
        mine = np.zeros((Ho, Wo))
        for dy, dx, wt in taps(Taps.get("conv", k, pad)):
            for a in range(Ho):
                for b in range(Wo):
                    iy, ix = a * stride + dy, b * stride + dx
                    if 0 <= iy < H and 0 <= ix < W:
                        mine[a, b] += x[iy, ix] * w.flat[wt]
        np.testing.assert_allclose(mine, y[0, 0].detach().numpy(), atol=1e-12)
        g = rng.standard_normal((Ho, Wo))
        y.backward(torch.from_numpy(g)[None, None])
        dx_ref = xt.grad[0, 0].numpy()
        dxm = np.zeros((H, W))
        classes = [(0, 0)] if stride == 1 else [(0, 0), (0, 1), (1, 0), (1, 1)]
        for py, px in classes:
            tl = Taps.get("dgrad1", k, pad) if stride == 1 else Taps.get("dgrad2", k, pad, py, px)
            for a in range((H - py + stride - 1) // stride):
                for b in range((W - px + stride - 1) // stride):
                    s = 0.0
                    for dy, dx, wt in taps(tl):
                        iy, ix = a + dy, b + dx
                        if 0 <= iy < Ho and 0 <= ix < Wo:
                            s += g[iy, ix] * w.flat[wt]
                    dxm[a * stride + py, b * stride + px] = s
        np.testing.assert_allclose(dxm, dx_ref, atol=1e-12)


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rpg_ramnet_amd.parallel import FlatGradReducer, shard_indices
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    red = FlatGradReducer(net, num_buckets=2)
    red.zero()
    data = torch.arange(8 * 3 * 6 * 6, dtype=torch.float32).reshape(8, 3, 6, 6) / 100.0
    mine = shard_indices(8, rank, world)
    net(data[mine]).square().mean().backward()
    assert all(p.grad.data_ptr() == red.views[p].data_ptr() for p in net.parameters())   # grads landed in the flat buffer
    red.all_reduce()
    red.wait()
    q.put((rank, mine, red.flat.clone().numpy()))
    dist.destroy_process_group()


def test_data_parallel_gradient_average_gloo_world2():
    """N>1 path on CPU (gloo, world_size 2): bucketed flat all-reduce == mean of the per-rank gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-6)
    # reference: single process, mean of the two shard gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    data = torch.arange(8 * 3 * 6 * 6, dtype=torch.float32).reshape(8, 3, 6, 6) / 100.0
    grads = []
    for idx in ([0, 2, 4, 6], [1, 3, 5, 7]):
        net.zero_grad()
        net(data[idx]).square().mean().backward()
        grads.append(torch.cat([p.grad.flatten() for p in reversed(list(net.parameters()))]))
    np.testing.assert_allclose(res[0][2], ((grads[0] + grads[1]) / 2).numpy(), rtol=1e-5, atol=1e-8)


def _dp8_worker(rank, world, port, q):
    """World-8 readiness, everything a node run relies on that does not need a device."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import loss_ref
    from rpg_ramnet_amd.data import ShardedSequenceSampler
    from rpg_ramnet_amd.parallel import FlatGradReducer, shard_indices
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 4, 3), torch.nn.Conv2d(4, 1, 1))
    red = FlatGradReducer(net, num_buckets=3)
    sent = []
    real = dist.all_reduce

    def spy(t, *a, **kw):                                # the order in which collectives leave this rank (sizes identify the buckets)
        sent.append(int(t.numel()))
        return real(t, *a, **kw)
    dist.all_reduce = spy
    n_seq = 67                                           # not a multiple of 8: the tail is dropped, every rank runs the same number of steps
    mine = shard_indices(n_seq, rank, world)
    samp = ShardedSequenceSampler(list(range(n_seq)), rank, world, shuffle=True, seed=3, batch_size=2)
    samp.set_epoch(5)
    picked = list(samp)
    g = torch.Generator().manual_seed(11)
    data = torch.randn(n_seq, 3, 8, 8, generator=g)
    target = torch.rand(n_seq, 1, 4, 4, generator=g)
    target[target < 0.1] = float("nan")                  # invalid pixels: the valid count differs between ranks
    # two backward passes per step (gradient accumulation), then ONE bucketed all-reduce
    red.zero()
    half = len(mine) // 2
    for idx in (mine[:half], mine[half:]):
        pred = torch.sigmoid(net(data[idx]))
        loss_ref.scale_invariant_loss(pred, target[idx]).backward()
    red.all_reduce()
    red.wait()
    # exact global-batch SI loss (trainer._sequence_loss_dp_exact's protocol): (sum d, sum d^2, n) all-reduced, loss from the global sums
    with torch.no_grad():
        pred = torch.sigmoid(net(data[mine]))
        d = (pred - target[mine]).double()
        ok = ~torch.isnan(d)
        table = torch.tensor([[float(d[ok].sum()), float((d[ok] ** 2).sum()), float(ok.sum()), 0.0]], dtype=torch.float64)
    dist.all_reduce = real
    dist.all_reduce(table)
    S1, S2, n = (float(v) for v in table[0, :3])
    q.put((rank, mine, picked, len(samp), sent, list(red.issue_log), red.flat.clone().numpy(), S2 / n - (S1 / n) ** 2))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world8_gloo():
    """SURVEY 8e at the world size of the node (8 ranks over gloo): rank r owns sequences r, r + 8, ...; the sharded sampler gives all
    ranks the same number of steps over disjoint items; every rank sends the same buckets in the same order; the bucketed all-reduce
    after TWO backward passes (gradient accumulation) is the mean over ranks of the accumulated gradients; the exact-loss table
    reproduces the single-process loss of the concatenated 8-shard batch (invalid pixels included)."""
    import torch.multiprocessing as mp
    from oracle import loss_ref
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_seq = 67
    for r, (rank, mine, picked, nsteps, sent, log, flat, loss) in enumerate(res):
        assert rank == r and mine == list(range(r, n_seq, world))
        assert nsteps == res[0][3] == (n_seq // (world * 2)) * 2 and len(picked) == nsteps
        assert sent == res[0][4] and len(sent) >= 2 and log == res[0][5] == [(0, len(sent))]        # same buckets, same order, on every rank
        np.testing.assert_allclose(flat, res[0][6], rtol=1e-6, atol=1e-9)                    # every rank holds the same average
        assert loss == res[0][7]
    allp = [i for t in res for i in t[2]]
    assert len(set(allp)) == len(allp)                                                       # the sampler's shards are disjoint
    # reference: ONE process — mean over the ranks of each rank's two accumulated passes; loss on the concatenated batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 4, 3), torch.nn.Conv2d(4, 1, 1))
    g = torch.Generator().manual_seed(11)
    data = torch.randn(n_seq, 3, 8, 8, generator=g)
    target = torch.rand(n_seq, 1, 4, 4, generator=g)
    target[target < 0.1] = float("nan")
    acc = None
    for r in range(world):
        mine = list(range(r, n_seq, world))
        half = len(mine) // 2
        net.zero_grad()
        for idx in (mine[:half], mine[half:]):
            loss_ref.scale_invariant_loss(torch.sigmoid(net(data[idx])), target[idx]).backward()
        gr = torch.cat([p.grad.flatten() for p in reversed(list(net.parameters()))])
        acc = gr.clone() if acc is None else acc + gr
    np.testing.assert_allclose(res[0][6], (acc / world).numpy(), rtol=2e-5, atol=1e-8)
    with torch.no_grad():
        order = [i for r in range(world) for i in range(r, n_seq, world)]
        want = float(loss_ref.scale_invariant_loss(torch.sigmoid(net(data[order])).double(), target[order].double()))
    assert abs(res[0][7] - want) < 1e-9 * max(1.0, abs(want))


def test_crop_parameters_match_the_reference():
    """Full-frame helper: ops.CropParameters reproduces every field of the reference's CropParameters (utils/inference_utils.py:287-314;
    fixture produced by the reference class itself) — the raw DAVIS frame 260 x 346 -> 264 x 352 with a (2, 2, 3, 3) reflection border
    — and the oracle run on a torch-reflect-padded frame and cropped with those fields gives the reference network's cropped output."""
    from rpg_ramnet_amd.ops import CropParameters
    from oracle import ramnet_ref
    z = load_golden("crop.npz")
    names = json.loads(str(z["field_names"]))
    for tag in ("davis", "odd", "odd2", "exact"):
        f = dict(zip(names, (int(v) for v in z[tag + ".fields"])))
        c = CropParameters(f["width"], f["height"], f["num_encoders"])
        assert {k: getattr(c, k) for k in names} == f, tag
        assert c.identity == (tag == "exact")
    f = dict(zip(names, (int(v) for v in z["davis.fields"])))
    assert (f["height_crop_size"], f["width_crop_size"], f["padding_top"], f["padding_left"]) == (264, 352, 2, 3)
    cfg = json.loads(str(z["net.config"]))
    from rpg_ramnet_amd.model import model as mm
    torch.manual_seed(0)
    sd = {k: v.detach() for k, v in mm.ERGB2DepthRecurrent(cfg).state_dict().items()}
    c = CropParameters(35, 26, 3)
    pad = torch.nn.ReflectionPad2d((c.padding_left, c.padding_right, c.padding_top, c.padding_bottom))
    prev, lstm = None, ramnet_ref.empty_states_lstm(2)
    for l in range(2):
        item = {k: pad(torch.from_numpy(z["net.item%d.%s" % (l, k)])) for k in ("events0", "events1", "image")}
        preds, supers, lstm = ramnet_ref.forward_recurrent(sd, cfg, item, prev, lstm)
        prev = supers["image"]
        for k, v in preds.items():
            np.testing.assert_allclose(c.crop(v).numpy(), z["net.pred%d.%s" % (l, k)], rtol=0, atol=2e-6)


def test_bench_plain_command_starts_its_own_ranks():
    """The driver's multi-GPU command is `python bench.py --gpus N ...` with no torch.distributed.run around it: bench.py must start
    its N ranks itself (rendezvous on 127.0.0.1), every rank must report in over the collective backend and rank 0 alone prints ONE
    JSON line.  --launch-check stops before any GPU work, so the launch contract runs here (gloo: no device in this container)."""
    import json as js
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = js.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks_seen"] == [0, 1]
    # ... and under torch.distributed.run (the other launch contract) the same file does not re-launch
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 100), os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert js.loads(lines[0])["rccl_ranks_seen"] == [0, 1]


def test_checkpoint_layout_round_trip_and_reference_pickle(tmp_path):
    """Checkpoint dict layout of base_trainer.py:142-150; a file that pickles the reference's `logger.logger.Logger`
    (not importable here) still loads, and its state_dict loads strict=True."""
    import pickle
    import sys
    import types
    from rpg_ramnet_amd import checkpoint as ck
    from rpg_ramnet_amd.model import model as mm
    cfg, z = ref_cfg("net_small_gru.npz")
    m = mm.ERGB2DepthRecurrent(cfg)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")})
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    path = ck.save_checkpoint(ck.checkpoint_name(str(tmp_path), 7, 0.1234), m, opt, 7, {"arch": "ERGB2DepthRecurrent", "model": cfg}, 0.5)
    assert path.endswith("checkpoint-epoch007-loss-0.1234.pth.tar")
    c = ck.load_checkpoint(path)
    assert set(c) == {"arch", "epoch", "logger", "state_dict", "optimizer", "monitor_best", "config"}
    assert c["arch"] == "ERGB2DepthRecurrent" and c["epoch"] == 7 and c["monitor_best"] == 0.5
    m2 = mm.ERGB2DepthRecurrent(cfg)
    start, _ = ck.resume(path, m2, torch.optim.Adam(m2.parameters(), lr=3e-4))
    assert start == 8
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # a checkpoint as the REFERENCE writes it: 'logger' is an instance of logger.logger.Logger
    pkg, sub = types.ModuleType("logger"), types.ModuleType("logger.logger")

    class Logger:                       # stand-in for the reference's class while pickling
        def __init__(self):
            self.entries = {1: {"epoch": 1, "loss": 0.5}}
    Logger.__module__, Logger.__qualname__ = "logger.logger", "Logger"
    sub.Logger = Logger
    sys.modules["logger"], sys.modules["logger.logger"] = pkg, sub
    try:
        ref_like = dict(c, logger=Logger())
        torch.save(ref_like, str(tmp_path / "ref.pth.tar"), pickle_module=pickle)
    finally:
        del sys.modules["logger"], sys.modules["logger.logger"]
    c2 = ck.load_checkpoint(str(tmp_path / "ref.pth.tar"))
    assert c2["logger"].entries[1]["loss"] == 0.5
    mm.ERGB2DepthRecurrent(cfg).load_state_dict(c2["state_dict"], strict=True)


def test_checkpoint_written_here_unpickles_without_this_package(tmp_path):
    """a checkpoint saved here must load with a plain torch.load in a process that has the reference's
    `logger.logger.Logger` but NOT rpg_ramnet_amd on sys.path (test.py / --resume of the reference)."""
    import subprocess
    import sys
    from rpg_ramnet_amd import checkpoint as ck
    net = torch.nn.Conv2d(1, 2, 3)
    path = ck.save_checkpoint(str(tmp_path / "c.pth.tar"), net, None, 3, {"arch": "X"})
    fake = tmp_path / "reftree" / "logger"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text("from .logger import Logger\n")
    (fake / "logger.py").write_text("class Logger:\n    def __init__(self):\n        self.entries = {}\n")
    code = ("import sys, torch; sys.path = [p for p in sys.path if 'repo' not in p]; sys.path.insert(0, %r); "
            "c = torch.load(%r, map_location='cpu', weights_only=False); "
            "assert 'rpg_ramnet_amd' not in sys.modules; "
            "assert type(c['logger']).__module__ == 'logger.logger' and c['logger'].entries == {}; print(c['epoch'])"
            % (str(tmp_path / "reftree"), path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip() == "3", r.stderr[-2000:]


def test_checkpoint_save_with_a_foreign_logger_package_already_imported(tmp_path):
    """the side-by-side setup — the reference tree's own `logger` package is imported before save_checkpoint().  Its
    sys.modules entries are overridden for the duration of the pickle and restored afterwards; an instance of the foreign
    Logger class handed in is re-wrapped (entries preserved)."""
    import sys
    import types
    from rpg_ramnet_amd import checkpoint as ck
    pkg, sub = types.ModuleType("logger"), types.ModuleType("logger.logger")

    class Logger:                       # the reference's class (RAM_Net/logger/logger.py): a different object of the same path
        def __init__(self):
            self.entries = {}
    Logger.__module__, Logger.__qualname__ = "logger.logger", "Logger"
    sub.Logger = pkg.Logger = Logger
    pkg.logger = sub
    sys.modules["logger"], sys.modules["logger.logger"] = pkg, sub
    try:
        net = torch.nn.Conv2d(1, 2, 3)
        foreign = Logger()
        foreign.entries[1] = {"epoch": 1, "loss": 0.25}
        p1 = ck.save_checkpoint(str(tmp_path / "a.pth.tar"), net, None, 1, {"arch": "X"})
        p2 = ck.save_checkpoint(str(tmp_path / "b.pth.tar"), net, None, 2, {"arch": "X"}, logger=foreign)
        assert sys.modules["logger"] is pkg and sys.modules["logger.logger"] is sub        # restored
        assert ck.load_checkpoint(p1)["logger"].entries == {}
        assert ck.load_checkpoint(p2)["logger"].entries[1]["loss"] == 0.25
        assert sys.modules["logger"] is pkg and sys.modules["logger.logger"] is sub
    finally:
        del sys.modules["logger"], sys.modules["logger.logger"]


def test_flat_reducer_adopts_grads_allocated_outside_the_flat_buffer():
    """optimizer.zero_grad(set_to_none=True) after zero() makes autograd allocate fresh .grad tensors; all_reduce()
    must fold them back into the flat buffer instead of averaging zeros."""
    from rpg_ramnet_amd.parallel import FlatGradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    red = FlatGradReducer(net)
    red.zero()
    net.zero_grad(set_to_none=True)
    net(torch.randn(2, 3, 6, 6)).square().mean().backward()
    assert any(p.grad.data_ptr() != red.views[p].data_ptr() for p in net.parameters())
    want = torch.cat([p.grad.flatten() for p in reversed(list(net.parameters()))]).clone()
    red.all_reduce()
    assert all(p.grad.data_ptr() == red.views[p].data_ptr() for p in net.parameters())
    np.testing.assert_allclose(red.flat.numpy(), want.numpy())


def test_cabi_argument_errors_and_host_only_entry_points():
    """Error behaviour of the C ABI without a GPU: bad arguments are rejected before any HIP call, with a message;
    size helpers are pure host functions."""
    from rpg_ramnet_amd import _hip
    L = _hip.lib()
    assert L.ramnet_conv_launch(None, None) == 10001 and b"bad argument" in L.ramnet_last_error()
    assert L.ramnet_wgrad_launch(None, None) == 10001
    d = _hip.ConvDesc()                      # all-zero descriptor: null pointers
    assert L.ramnet_conv_launch(ctypes.byref(d), None) == 10001
    assert L.ramnet_si_loss_fwd(None, None, 0, 1.0, 1.0, None, None, None) == 10001
    # a source tensor beyond the 32-bit element offsets of the patch loaders is refused on the host (dummy non-null pointers:
    # the check comes before any HIP call)
    d = _hip.ConvDesc()
    d.x0 = d.w = d.out = 4096
    d.ntaps, d.stride, d.B, d.Ho, d.Wo, d.Hin, d.Win, d.Cout = 9, 1, 4096, 64, 64, 4096, 4096, 64
    d.C0, d.ld0, d.ldo, d.osy, d.osx = 64, 64, 64, 1, 1
    d.algo = _hip.ALGO_WINOGRAD
    assert L.ramnet_conv_launch(ctypes.byref(d), None) == 10001 and b"px * ld" in L.ramnet_last_error()
    assert L.ramnet_voxelize(None, 10, 0, 4, 4, None, None) == 10001
    # packed sizes: [tap][chunk16][Cout_pad32][16]
    assert L.ramnet_packed_weight_elems(64, 32, 5, 5, 0, 1) == 25 * 2 * 64 * 16
    assert L.ramnet_packed_weight_elems(64, 32, 5, 5, 1, 1) == 25 * 4 * 32 * 16          # transposed: reduce over O=64
    assert L.ramnet_packed_weight_elems(256, 128, 3, 3, 0, 4) == 9 * 8 * 256 * 16         # LSTM gates: 4 x pad32(64)
    assert L.ramnet_packed_weight_elems(32, 5, 5, 5, 0, 1) == 25 * 1 * 32 * 16            # 5 input channels -> one chunk
    assert L.ramnet_msg_workspace_elems(2, 32, 48, 4) == 2 * (32 * 48 + 16 * 24 + 8 * 12 + 4 * 6)
    assert L.ramnet_msg_workspace_elems(2, 4, 4, 4) == 0                                   # 8x pooling of a 4x4 map


def test_folded_upsample_conv_algebra_cpu():
    """conv5x5_zero_padded(bilinear_x2(x)) == four 4x4 parity convolutions of the replicate-padded input (ops.fold_weights)
    + the border GEMMs (tr.border_matrices): the factorisation the HIP path runs, checked in float64 with torch ops only."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(0)
    B, Cin, Cout, H, W = 2, 3, 5, 6, 7
    x = torch.randn(B, Cin, H, W, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 5, 5, dtype=torch.float64)
    u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(u, w, None, 1, 2)
    w4 = ops.fold_weights(w)                                          # [O][I][py][px][ty][tx]
    xp = F.pad(x, (2, 2, 2, 2), mode="replicate")
    y = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            full = F.conv2d(xp, w4[:, :, py, px])                     # valid 4x4 conv: (H+1) x (W+1) positions
            y[:, :, py::2, px::2] = full[:, :, py:py + H, px:px + W]   # tap t reads padded row i + py + t
    H2, W2 = 2 * H, 2 * W
    assert torch.allclose(y[:, :, 2:H2 - 2, 2:W2 - 2], ref[:, :, 2:H2 - 2, 2:W2 - 2], atol=1e-12)          # interior: exact as is
    wr, wc = tr.border_matrices(w)
    idx = torch.arange(W2)
    for side, row in enumerate((0, H2 - 1)):                          # rows: A[(b, ox)][(kx, ci)] = u[row][clamp(ox + kx - 2)]
        A = torch.stack([u[:, :, row, (idx + kx - 2).clamp(0, W2 - 1)] for kx in range(5)], 1)              # [B][5][Cin][W2]
        G = (A.permute(0, 3, 1, 2).reshape(B * W2, 5 * Cin) @ wr[side]).view(B, W2, 2, Cout)
        for slot in range(2):
            y[:, :, (H2 - 2 + slot) if side else slot, :] += G[:, :, slot].permute(0, 2, 1)
    idy = torch.arange(H2)
    for side, col in enumerate((0, W2 - 1)):                          # columns: rows outside the image contribute nothing
        cols = []
        for ky in range(5):
            r = idy + ky - 2
            v = u[:, :, r.clamp(0, H2 - 1), col] * ((r >= 0) & (r < H2)).to(u.dtype)
            cols.append(v)
        A = torch.stack(cols, 1)                                      # [B][5][Cin][H2]
        G = (A.permute(0, 3, 1, 2).reshape(B * H2, 5 * Cin) @ wc[side]).view(B, H2, 2, Cout)
        for slot in range(2):
            y[:, :, :, (W2 - 2 + slot) if side else slot] += G[:, :, slot].permute(0, 2, 1)
    assert torch.allclose(y, ref, atol=1e-12), float((y - ref).abs().max())


def test_winograd_f2x2_3x3_algebra_cpu():
    """The transforms hard-wired in csrc/conv_wino.hip / conv_wgrad_wino.hip, in float64 torch: forward
    Y = A^T[(G g G^T) .* (B^T d B)]A per 4x4 input tile, and backward-weights dg = G^T [sum_tiles (B^T d B) .* (A dy A^T)] G,
    against F.conv2d and its autograd."""
    import torch.nn.functional as F
    torch.manual_seed(1)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    Cin, Cout, H, W = 3, 4, 6, 8
    x = torch.randn(1, Cin, H, W, dtype=torch.float64)
    g = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(x, g, None, 1, 1)
    dy = torch.randn_like(ref)
    (ref * dy).sum().backward()
    xp = F.pad(x, (1, 1, 1, 1))
    U = torch.einsum("ia,ocab,jb->ocij", G, g.detach(), G)                               # [co][ci][4][4]
    y = torch.zeros_like(ref)
    dU = torch.zeros_like(U)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[0, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                           # [ci][4][4]
            V = torch.einsum("ia,cab,jb->cij", Bt, d, Bt)
            M = torch.einsum("ocij,cij->oij", U, V)
            y[0, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum("ai,oij,bj->oab", At, M, At)
            Z = torch.einsum("ia,oab,jb->oij", At.t(), dy[0, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2], At.t())
            dU += torch.einsum("cij,oij->ocij", V, Z)
    assert torch.allclose(y, ref, atol=1e-12)
    dg = torch.einsum("ia,ocij,jb->ocab", G, dU, G)                                      # G^T dU G
    assert torch.allclose(dg, g.grad, atol=1e-12)


def test_space_to_depth_weight_maps_cpu():
    """conv5x5 stride 2 == conv3x3 stride 1 of the space-to-depth input with ops.s2d_weights; its adjoint routes the gradient
    of the 3x3 / 4*Cin weights back to the 5x5 ones (what S2DConvParam.finalize does)."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(2)
    B, Cin, Cout, H, W = 2, 3, 4, 8, 10
    x = torch.randn(B, Cin, H, W, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 5, 5, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(x, w, None, 2, 2)
    xs = F.pixel_unshuffle(x, 2).view(B, Cin, 4, H // 2, W // 2).permute(0, 2, 1, 3, 4).reshape(B, 4 * Cin, H // 2, W // 2)
    w3 = ops.s2d_weights(w.detach()).requires_grad_(True)
    y = F.conv2d(xs, w3, None, 1, 1)
    assert torch.allclose(y, ref, atol=1e-12)
    assert int((ops.s2d_weights(torch.ones(1, 1, 5, 5)) != 0).sum()) == 25          # 25 of the 36 slices are populated
    g = torch.randn_like(ref)
    (ref * g).sum().backward()
    (y * g).sum().backward()
    assert torch.allclose(ops.s2d_weights_adjoint(w3.grad, Cin), w.grad, atol=1e-12)


def test_winograd_f2x2_4x4_algebra_cpu():
    """The matrices of csrc/conv_wino24.hip / tr.fold_weights_wino: Y = A^T[(G g G^T) .* (B^T d B)]A equals the 4x4 stride-1
    correlation of every parity class of the folded upsample-conv, and the four classes together equal the 5x5 convolution of the
    bilinear upsample away from the border."""
    import torch.nn.functional as F
    from rpg_ramnet_amd import ops
    torch.manual_seed(0)
    Cin, Cout, Hh, W = 3, 2, 6, 8
    w = torch.randn(Cout, Cin, 5, 5, dtype=torch.float64)
    x = torch.randn(1, Cin, Hh, W, dtype=torch.float64)
    xpad = F.pad(x, (2, 2, 2, 2), mode="replicate")
    U = tr.fold_weights_wino(w)                                   # [4][25][Cin][Cout]
    BT = torch.tensor(tr.W24_BT, dtype=torch.float64)
    AT = torch.tensor(tr.W24_AT, dtype=torch.float64)
    W4 = ops.fold_weights(w)                                       # [O][I][py][px][4][4]
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, w, None, 1, 2)
    y = torch.zeros(1, Cout, 2 * Hh, 2 * W, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            cls = py * 2 + px
            direct = F.conv2d(xpad[:, :, py:py + Hh + 3, px:px + W + 3], W4[:, :, py, px])          # [1][Cout][Hh][W]
            for ty in range(Hh // 2):
                for tx in range(W // 2):
                    d = xpad[0, :, 2 * ty + py:2 * ty + py + 5, 2 * tx + px:2 * tx + px + 5]        # [Cin][5][5]
                    v = torch.einsum("ar,crs,bs->cab", BT, d, BT).reshape(Cin, 25)
                    m = torch.einsum("cp,pco->po", v, U[cls]).reshape(5, 5, Cout)
                    out = torch.einsum("ai,ijo,bj->oab", AT, m, AT)                               # [Cout][2][2]
                    assert torch.allclose(out, direct[0, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2], atol=1e-10)
                    y[0, :, 4 * ty + py:4 * ty + py + 4:2, 4 * tx + px:4 * tx + px + 4:2] = out
    assert torch.allclose(y[:, :, 2:-2, 2:-2], ref[:, :, 2:-2, 2:-2], atol=1e-10)
    # packed layout: element (cls, pos, k, n) sits where include/ramnet_hip.h says
    w2 = torch.randn(64, 32, 5, 5)
    packed, U2 = tr.pack_fold_wino(w2), tr.fold_weights_wino(w2).float()
    for cls, pos, k, n in [(0, 0, 0, 0), (3, 24, 31, 63), (2, 7, 21, 37), (1, 13, 6, 50)]:
        idx = ((((cls * 2 + k // 16) * 1 + n // 64) * 25 + pos) * 4 + (n % 64) // 16) * 256 + (((k % 16) // 4) * 16 + n % 16) * 4 + k % 4
        assert packed[idx] == U2[cls, pos, k, n]
    w4 = torch.randn(32, 64, 5, 5)                                 # 32-channel layer, Cin % 32 == 0: pair layout — class = row parity,
    packed, U4 = tr.pack_fold_wino(w4), tr.fold_weights_wino(w4).float()      # column = (column parity, channel), chunks of 16
    for cls, pos, k, n in [(0, 0, 0, 0), (3, 24, 63, 31), (2, 7, 21, 17), (1, 13, 38, 5)]:
        py, col = cls >> 1, (cls & 1) * 32 + n
        idx = (((py * 4 + k // 16) * 25 + pos) * 4 + col // 16) * 256 + (((k % 16) // 4) * 16 + col % 16) * 4 + k % 4
        assert packed[idx] == U4[cls, pos, k, n]
    w3 = torch.randn(32, 24, 5, 5)                                 # 32-channel layers: chunks of 8, two 16-channel groups
    packed, U3 = tr.pack_fold_wino(w3), tr.fold_weights_wino(w3).float()
    for cls, pos, k, n in [(0, 0, 0, 0), (3, 24, 23, 31), (2, 7, 13, 17)]:
        idx = ((((cls * 3 + k // 8) * 1 + n // 32) * 25 + pos) * 2 + (n % 32) // 16) * 128 + (((k % 8) // 2) * 16 + n % 16) * 2 + k % 2
        assert packed[idx] == U3[cls, pos, k, n]


def test_epoch_trainer_matches_the_reference_base_trainer_fixture(tmp_path):
    """SURVEY 8f-2, epoch level: rpg_ramnet_amd.trainer.EpochTrainer against tests/golden/trainer.json, which was produced by
    RUNNING the reference's BaseTrainer (base_trainer.py:36-43, 65-123, 133-158) with the same scripted epoch function
    (tests/golden/make_golden_trainer.py): learning-rate sequence (ExponentialLR stepped every lr_scheduler_freq epochs),
    monitor_best sequence, the checkpoint files left behind (periodic + model_best) and the checkpoint dict."""
    import json
    import os
    from rpg_ramnet_amd import checkpoint as ck
    from rpg_ramnet_amd.trainer import EpochTrainer
    z = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer.json")))
    cfg = json.loads(json.dumps(z["config"]))
    cfg["trainer"]["save_dir"] = str(tmp_path)
    torch.manual_seed(0)
    net = torch.nn.Conv2d(1, 2, 3)
    lrs, mbs = [], []

    def epoch_fn(epoch):
        lrs.append(t.optimizer.param_groups[0]["lr"])
        mbs.append(t.monitor_best)
        loss = z["losses"][epoch - 1]
        return {"loss": loss, "val_loss": loss + 0.1 * (-1) ** epoch}

    t = EpochTrainer(net, cfg, epoch_fn)
    t.train()
    np.testing.assert_allclose(lrs, z["lr_at_epoch_start"], rtol=1e-12)
    assert t.optimizer.param_groups[0]["lr"] == pytest.approx(z["lr_after_last_epoch"], rel=1e-12)
    assert [m if m != float("inf") else "inf" for m in mbs] == z["monitor_best_at_epoch_start"]
    assert t.monitor_best == z["monitor_best_final"]
    d = os.path.join(str(tmp_path), cfg["name"])
    assert sorted(f for f in os.listdir(d) if f.endswith(".pth.tar")) == z["files"]
    c = ck.load_checkpoint(os.path.join(d, "model_best.pth.tar"))
    assert sorted(c.keys()) == z["checkpoint_keys"]
    assert c["epoch"] == z["best"]["epoch"] and c["monitor_best"] == z["best"]["monitor_best"] and c["arch"] == z["best"]["arch"]
    assert len(c["logger"].entries) == z["best"]["logger_entries"] and sorted(c["optimizer"].keys()) == z["best"]["optimizer_keys"]
    # resume (base_trainer.py:160-179): continues at epoch + 1 with the stored monitor_best and optimizer state
    cfg["trainer"]["epochs"] = 8
    t2 = EpochTrainer(torch.nn.Conv2d(1, 2, 3), cfg, lambda e: {"loss": 1.0, "val_loss": 1.0}, resume=os.path.join(d, "model_best.pth.tar"))
    assert t2.start_epoch == 8 and t2.monitor_best == z["monitor_best_final"]
    t2.train()
    assert len(t2.train_logger.entries) == 8


def test_time_split_join_autograd_plumbing_cpu():
    """Host logic of the time-batched forward (model._forward_time_batched; no kernels): ops.TimeSplit hands out views of a batched
    tensor and concatenates the gradients, ops.TimeJoin declares cell outputs that already ARE consecutive slots of one buffer to be
    its [n B, ...] view (no copy; refuses anything else), model._cat_batch turns adjacent tensors into a view and copies otherwise."""
    from rpg_ramnet_amd import ops
    from rpg_ramnet_amd.model.model import _cat_batch

    class Cell(torch.autograd.Function):          # stand-in for GRUCell(out=): writes the slot through a raw pointer (no version bump)
        @staticmethod
        def forward(ctx, x, out):
            out.numpy()[...] = (2.0 * x).numpy()
            return out

        @staticmethod
        def backward(ctx, g):
            return 2.0 * g, None

    n, B = 3, 2
    x = torch.randn(n * B, 4, 5, 8, requires_grad=True)
    parts = ops.TimeSplit.apply(x * 1.0, n)
    assert all(p.shape == (B, 4, 5, 8) for p in parts)
    arena = torch.empty(n, B, 4, 5, 8)
    hs = [Cell.apply(p, arena[k]) for k, p in enumerate(parts)]
    assert all(h.data_ptr() == arena[k].data_ptr() and h.requires_grad for k, h in enumerate(hs))
    joined = ops.TimeJoin.apply(arena[0:n].view(n * B, 4, 5, 8), *hs)
    assert joined.data_ptr() == arena.data_ptr() and joined.requires_grad
    w = torch.randn(n * B, 4, 5, 8)
    (joined * w).sum().backward()
    torch.testing.assert_close(x.grad, 2.0 * w)
    with pytest.raises(RuntimeError):             # parts that are not the buffer's slots, in order
        ops.TimeJoin.apply(arena[0:2].view(2 * B, 4, 5, 8), hs[1].detach(), hs[0].detach())
    g = torch.randn(5, B, 3, 4, 6)
    a = _cat_batch([g[k] for k in range(5)])
    assert a.data_ptr() == g.data_ptr() and torch.equal(a, g.view(5 * B, 3, 4, 6))
    b = _cat_batch([g[1], g[3]])
    assert b.data_ptr() != g[1].data_ptr() and torch.equal(b, torch.cat([g[1], g[3]], 0))
    c = _cat_batch([g[2].clone(), g[3].clone()])
    assert torch.equal(c, g[2:4].reshape(2 * B, 3, 4, 6))


def test_time_fan_autograd_plumbing_cpu():
    """ops.TimeFan (a batched feature that feeds the next layer of the chain as a whole AND n cells slice by slice): the gradient is
    the batched consumer's plus the concatenation of the slices' — with unused slices, and with an unused batched consumer."""
    from rpg_ramnet_amd import ops
    n, B = 3, 2
    x = torch.randn(n * B, 4, 5, 8, requires_grad=True)
    w_all, w = torch.randn(n * B, 4, 5, 8), [torch.randn(B, 4, 5, 8) for _ in range(n)]
    out = ops.TimeFan.apply(x * 1.0, n)
    assert out[0].shape == x.shape and all(p.shape == (B, 4, 5, 8) for p in out[1:])
    ((out[0] * w_all).sum() + sum((p * wk).sum() for p, wk in zip(out[1:], w))).backward()
    torch.testing.assert_close(x.grad, w_all + torch.cat(w, 0))
    x.grad = None
    out = ops.TimeFan.apply(x * 1.0, n)
    ((out[0] * w_all).sum() + (out[2] * w[1]).sum()).backward()                # slices 0 and 2 unused
    torch.testing.assert_close(x.grad, w_all + torch.cat([torch.zeros_like(w[0]), w[1], torch.zeros_like(w[2])], 0))
    x.grad = None
    out = ops.TimeFan.apply(x * 1.0, n)
    sum((p * wk).sum() for p, wk in zip(out[1:], w)).backward()                # the batched consumer unused
    torch.testing.assert_close(x.grad, torch.cat(w, 0))


def test_time_fan_relu_premask_cpu():
    """The fan-in of a time-batched ReLU feature applies the feature's own ReLU mask (premask=True): gradient = (sum) * (x > 0), on every path
    of TimeFan / TimeSplit (all consumers, slices only, batched consumer only); ops.premask_relu_feature only says yes for the output of a
    ConvAct-like node that allows it."""
    from rpg_ramnet_amd import ops
    n, B = 3, 2
    torch.manual_seed(3)
    x = torch.randn(n * B, 4, 5, 8, requires_grad=True)
    w_all, w = torch.randn(n * B, 4, 5, 8), [torch.randn(B, 4, 5, 8) for _ in range(n)]
    y = torch.relu(x).contiguous()
    m = (y > 0).float()
    out = ops.TimeFan.apply(y, n, True)
    ((out[0] * w_all).sum() + sum((p * wk).sum() for p, wk in zip(out[1:], w))).backward()
    torch.testing.assert_close(x.grad, (w_all + torch.cat(w, 0)) * m * m)        # (torch.relu's own backward masks once more: idempotent)
    # the masked sum itself, without a second mask behind it
    z = (x.detach() * 1.0).requires_grad_(True)
    zz = z * 1.0
    out = ops.TimeFan.apply(zz, n, True)
    ((out[0] * w_all).sum() + sum((p * wk).sum() for p, wk in zip(out[1:], w))).backward()
    torch.testing.assert_close(z.grad, (w_all + torch.cat(w, 0)) * (z.detach() > 0).float())
    z.grad = None
    out = ops.TimeFan.apply(z * 1.0, n, True)
    (out[0] * w_all).sum().backward()                                            # the batched consumer alone
    torch.testing.assert_close(z.grad, w_all * (z.detach() > 0).float())
    z.grad = None
    parts = ops.TimeSplit.apply(z * 1.0, n, True)
    sum((p * wk).sum() for p, wk in zip(parts, w)).backward()
    torch.testing.assert_close(z.grad, torch.cat(w, 0) * (z.detach() > 0).float())
    assert not ops.premask_relu_feature(torch.relu(x))                           # not a node that declared premask_ok
    assert not ops.premask_relu_feature(x.detach())


def _bf16_rne(x):
    """float32 tensor -> the nearest bf16 value (ties to even), as float32: what v_cvt_pk_bf16_f32 / the pack kernel's bf16_rne produce."""
    u = x.contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return (u - ((u >> 31) << 32)).to(torch.int32).view(torch.float32)


def _split3(x):
    h = _bf16_rne(x)
    r1 = x - h
    m = _bf16_rne(r1)
    r2 = r1 - m
    return h, m, _bf16_rne(r2), r2


def test_split_operand_winograd_f2x4_algebra_cpu():
    """The arithmetic hard-wired in csrc/conv_wino6s.hip, restated in torch on the CPU: F(2,3) down the rows x F(4,3) along the columns
    (Y = A2^T [(G2 g G4^T) .* (B2^T d B4)] A4, the matrices of csrc/conv_wino6.hip) with BOTH Winograd-domain operands as three-term
    bf16 splits (round-to-nearest terms, exact fp32 residuals) and six of the nine partial products accumulated in float32 — against
    float64 F.conv2d: the split form must be as close to float64 as the plain fp32 evaluation of the same transform (the dropped products
    are <= 2^-25 of a product), the residuals must be exact, and three terms must reproduce an fp32 value to <= 2^-24 relative."""
    import torch.nn.functional as F
    torch.manual_seed(2)
    G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                      dtype=torch.float64)
    B2t = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    B4t = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                       dtype=torch.float64)
    A2t = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    A4t = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    Cin, Cout, H, W = 48, 8, 4, 8
    x = torch.randn(1, Cin, H, W)
    g = torch.randn(Cout, Cin, 3, 3) * 0.1
    ref = F.conv2d(x.double(), g.double(), None, 1, 1)
    # the transform algebra itself, in float64
    xp = F.pad(x.double(), (1, 1, 1, 1))
    U64 = torch.einsum("ia,ocab,jb->ocij", G2, g.double(), G4)                             # [co][ci][4][6]
    U = U64.float()                                                                        # (the packs round U to fp32 first)
    y64, y32, ysp = torch.zeros_like(ref), torch.zeros(ref.shape), torch.zeros(ref.shape)
    Uh, Um, Ul, _ = _split3(U)
    for ty in range(H // 2):
        for tx in range(W // 4):
            d = xp[0, :, 2 * ty:2 * ty + 4, 4 * tx:4 * tx + 6]
            V64 = torch.einsum("ia,cab,jb->cij", B2t, d, B4t)                              # [ci][4][6]
            y64[0, :, 2 * ty:2 * ty + 2, 4 * tx:4 * tx + 4] = torch.einsum("ai,oij,bj->oab", A2t, torch.einsum("ocij,cij->oij", U64, V64), A4t)
            V = torch.einsum("ia,cab,jb->cij", B2t.float(), d.float(), B4t.float())        # fp32 transform (small integer coefficients)
            M32 = torch.einsum("ocij,cij->oij", U, V)
            Vh, Vm, Vl, r2 = _split3(V)
            assert torch.equal((Vh.double() + Vm.double() + r2.double()).float(), V), "residuals are exact"
            assert float(((Vh.double() + Vm.double() + Vl.double()) - V.double()).abs().max() / V.abs().max()) <= 2.0 ** -24
            Msp = sum(torch.einsum("ocij,cij->oij", ub, va) for va, ub in ((Vl, Uh), (Vm, Um), (Vh, Ul), (Vm, Uh), (Vh, Um), (Vh, Uh)))
            for M, y in ((M32, y32), (Msp, ysp)):
                y[0, :, 2 * ty:2 * ty + 2, 4 * tx:4 * tx + 4] = torch.einsum("ai,oij,bj->oab", A2t.float(), M, A4t.float())
    assert torch.allclose(y64, ref, atol=1e-12), "F(2x4,3x3) reproduces the convolution"
    e32 = float((y32.double() - ref).abs().max() / ref.abs().max())
    esp = float((ysp.double() - ref).abs().max() / ref.abs().max())
    assert e32 < 1e-5 and esp < 1e-5 and esp <= 1.5 * e32 + 1e-7, (e32, esp)
    # the pack's size entry point: [chunk of 16][64-channel block][24 positions][2 halves][3 planes][64 lanes x 16 bytes], in 4-byte units
    from rpg_ramnet_amd import _hip
    L = _hip.lib()
    assert L.ramnet_packed_weight_elems_wino2x4_split(128, 128, 0) == 8 * 2 * 24 * 2 * 3 * 256
    assert L.ramnet_packed_weight_elems_wino2x4_split(128, 40, 0) == 3 * 2 * 24 * 2 * 3 * 256          # ragged reduction depth: 3 chunks
    assert L.ramnet_packed_weight_elems_wino2x4_split(128, 64, 1) == 8 * 1 * 24 * 2 * 3 * 256          # backward-data: reduce over Cout
    assert L.ramnet_conv_wino_split_ok(None, 0) == 0


def test_arena_slots_are_aliases_with_version_counters_of_their_own():
    """ops.arena_slots (the state slots the cells of the time-batched forward write, GRUCell / LSTMCell `out=`): consecutive slots of ONE
    buffer that are not autograd views of it — an in-place write to one slot (what ctx.mark_dirty declares for a cell's output) moves only
    that slot's version counter, so what a backward saved of the neighbouring slots stays valid and a rewrite of the SAME slot is caught."""
    from rpg_ramnet_amd import ops
    buf, slots = ops.arena_slots(3, (2, 4, 5, 8), torch.device("cpu"))
    assert buf.shape == (3, 2, 4, 5, 8) and len(slots) == 3
    for k, s in enumerate(slots):
        assert s.data_ptr() == buf[k].data_ptr() and s.is_contiguous() and not s._is_view() and tuple(s.shape) == (2, 4, 5, 8)
    v = [s._version for s in slots]
    slots[1].fill_(7.0)
    assert slots[1]._version == v[1] + 1 and slots[0]._version == v[0] and slots[2]._version == v[2] and buf._version == 0
    assert float(buf[1].min()) == 7.0 and float(buf[1].max()) == 7.0        # (the write landed in the shared storage)

    class Write(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, out):
            out.copy_(2 * x.detach())
            ctx.mark_dirty(out)
            return out

        @staticmethod
        def backward(ctx, g):
            return 2 * g, None

    class Keep(torch.autograd.Function):
        @staticmethod
        def forward(ctx, h):
            ctx.save_for_backward(h)
            return h * 3

        @staticmethod
        def backward(ctx, g):
            (h,) = ctx.saved_tensors
            return g * 3

    x = torch.ones(2, 4, 5, 8, requires_grad=True)
    _, slots = ops.arena_slots(2, (2, 4, 5, 8), torch.device("cpu"))
    z = Keep.apply(Write.apply(x, slots[0]))
    Write.apply(x, slots[1])                                                # another slot: harmless
    z.sum().backward()
    assert float(x.grad.min()) == 6.0
    _, slots = ops.arena_slots(2, (2, 4, 5, 8), torch.device("cpu"))
    z = Keep.apply(Write.apply(x, slots[0]))
    Write.apply(x, slots[0])                                                # the same slot again: the saved state is gone
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        z.sum().backward()


def test_bench_traffic_is_requests_times_64_bytes_plus_writes():
    """bench.py's roofline.traffic: FETCH_SIZE + WRITE_SIZE as counted (= TCC_EA_RDREQ x 64 B + WRITE_SIZE), launch-weighted
    over a kernel's grids; the doubled-FETCH figure only on request."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    pmc = {"conv_wino_r6_kernel<4,0>": {"720896": {"fetch_kb_raw": 100.0, "write_kb": 50.0, "hbm_bytes_per_launch": 250.0 * 1024, "launches": 3},
                                        "196608": {"fetch_kb_raw": 40.0, "write_kb": 10.0, "hbm_bytes_per_launch": 90.0 * 1024, "launches": 1}},
           "other_kernel": {"1": {"fetch_kb_raw": 1.0, "write_kb": 1.0, "hbm_bytes_per_launch": 3.0 * 1024, "launches": 7}}}
    assert bench.traffic_of(pmc, "conv_wino_r6_kernel<4,0>") == pytest.approx((3 * 150.0 + 50.0) * 1024 / 4)
    assert bench.traffic_of(pmc, "conv_wino_r6_kernel<4,0>", fetch_x2=True) == pytest.approx((3 * 250.0 + 90.0) * 1024 / 4)
    assert bench.traffic_of(pmc, "no_such_kernel") is None
    assert bench.pmc_launches_of(pmc, "conv_wino_r6_kernel<4,0>") == 4
