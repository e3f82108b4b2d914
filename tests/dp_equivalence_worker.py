"""Worker of tests/test_hip_model.py::test_data_parallel_hip_model_gradient_equivalence (launched by torch.distributed.run,
2 ranks, gloo, both ranks on cuda:0): data-parallel gradients of the HIP model == mean of the single-rank gradients."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]


def main():
    from recipe import make_item
    from util import build_hip_model, ref_cfg
    from rpg_ramnet_amd.parallel import FlatGradReducer, shard_indices
    from rpg_ramnet_amd.trainer import sequence_loss
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    cfg, _ = ref_cfg("net_seeded_ramnet.npz", every_x_rgb_frame=2, loss_composition=["image", "events1"])
    model = build_hip_model("ERGB2DepthRecurrent", cfg).train()          # torch.manual_seed(0): identical weights on every rank
    red = FlatGradReducer(model, overlap="--overlap" in sys.argv)      # (the plain mode reads the LOCAL gradients between backward and all_reduce)
    rng = np.random.default_rng(21)
    n_seq, L = 4, 2                                                      # 4 sequences of 2 packages, sharded r, r + world, ...
    data = [[make_item(rng, 1, 32, 48, 2, 5, 1, True, 0.1) for _ in range(L)] for _ in range(n_seq)]

    def batch(idx):
        return [{k: torch.cat([data[i][l][k] for i in idx]) for k in data[0][l]} for l in range(L)]

    def local_grads(idx):
        red.zero(arm=False)
        total, _ = sequence_loss(model, batch(idx), cfg["loss_composition"], [1, 1])
        total.backward()
        torch.cuda.synchronize()
        return red.flat.clone()

    mine = shard_indices(n_seq, rank, world)
    if "--exact" in sys.argv:
        return exact_mode(model, red, cfg, batch, mine, n_seq, rank, world)
    if "--accum" in sys.argv:
        return accum_mode(model, red, cfg, batch, mine, rank)
    own = local_grads(mine)
    assert all(p.grad.data_ptr() == red.views[p].data_ptr() for p in model.parameters())
    red.all_reduce()
    red.wait()
    torch.cuda.synchronize()
    avg = red.flat.clone()
    seen = [None] * world
    dist.all_gather_object(seen, (rank, mine))
    gathered = [torch.empty_like(own) for _ in range(world)]
    dist.all_gather(gathered, own)
    out = {"rank": rank, "ranks_seen": seen}
    if rank == 0:
        # what ONE process computes for each shard alone, and their mean
        single = [local_grads(shard_indices(n_seq, r, world)) for r in range(world)]
        mean = sum(single) / world
        scale = float(mean.abs().max())
        out.update(err_vs_single_rank_mean=float((avg - mean).abs().max()) / scale,
                   err_vs_gathered_mean=float((avg - sum(gathered) / world).abs().max()) / scale,
                   shards_differ=float((single[0] - single[1]).abs().max()) / scale, n=int(avg.numel()))
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def exact_mode(model, red, cfg, batch, mine, n_seq, rank, world):
    """dp_exact: SI statistics all-reduced before the backward -> averaged gradient == gradient of ONE process on the concatenated
    batch (model/loss.py:9: mean(d)^2 over the whole batch), checked against the HIP model itself and against the fp64 oracle."""
    from rpg_ramnet_amd.trainer import sequence_loss
    lc = cfg["loss_composition"]
    red.zero()
    total, reported = sequence_loss(model, batch(mine), lc, [1, 1], dp_exact=True)
    total.backward()
    red.all_reduce()
    red.wait()
    torch.cuda.synchronize()
    avg, loss_dp = red.flat.clone(), float(total.detach())
    # the standard-DDP gradient of the same shards, for contrast
    red.zero()
    t2, _ = sequence_loss(model, batch(mine), lc, [1, 1])
    t2.backward()
    red.all_reduce()
    red.wait()
    torch.cuda.synchronize()
    ddp = red.flat.clone()
    losses = [None] * world
    dist.all_gather_object(losses, loss_dp)
    if rank == 0:
        full = [i for r in range(world) for i in range(r, n_seq, world)]      # concatenation in rank order (order is irrelevant to the loss)
        red.zero(arm=False)                                                    # (rank 0 alone: no collective may leave during this backward)
        t1, _ = sequence_loss(model, batch(full), lc, [1, 1])
        t1.backward()
        torch.cuda.synchronize()
        single = red.flat.clone()
        from oracle import ramnet_ref
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
        seq64 = [{k: v.double().cpu() for k, v in it.items()} for it in batch(full)]
        ref, _ = ramnet_ref.sequence_loss(sd, cfg, seq64, lc, [1, 1])
        ref.backward()
        gmax = max(float(v.grad.abs().max()) for v in sd.values())
        worst = 0.0
        for k, p in model.named_parameters():
            g = avg[_offset(red, p):_offset(red, p) + p.numel()].view_as(p).cpu().double()
            worst = max(worst, float((g - sd[k].grad).abs().max() / max(float(sd[k].grad.abs().max()), 1e-2 * gmax)))
        scale = float(single.abs().max())
        print(json.dumps({"rank": rank, "losses": losses, "loss_single": float(t1.detach()), "loss_oracle": float(ref.detach()),
                          "err_vs_single_rank_concat": float((avg - single).abs().max()) / scale,
                          "ddp_vs_single_rank_concat": float((ddp - single).abs().max()) / scale,
                          "err_vs_oracle_concat": worst, "n": int(avg.numel()), "early_buckets": red.early_buckets,
                          "buckets": len(red.buckets)}))
    dist.barrier()
    dist.destroy_process_group()


def accum_mode(model, red, cfg, batch, mine, rank):
    """two backward passes per optimizer step with the early bucket issue on.  The armed first pass sends buckets from the
    fold; the second pass must neither race with them nor leave its share of those buckets un-reduced."""
    from rpg_ramnet_amd.trainer import sequence_loss
    lc = cfg["loss_composition"]
    halves = [mine[:len(mine) // 2], mine[len(mine) // 2:]]

    def run(arm_first, arm_last):
        red.zero(arm=arm_first)
        for j, idx in enumerate(halves):
            if arm_last and j == len(halves) - 1:
                red.arm()
            total, _ = sequence_loss(model, batch(idx), lc, [1, 1])
            total.backward()
        red.all_reduce()
        red.wait()
        torch.cuda.synchronize()
        return red.flat.clone()

    e0 = red.early_buckets
    got_first = run(True, False)          # zero() arms the FIRST pass (the pattern the advisor flagged)
    e1 = red.early_buckets
    got_last = run(False, True)           # the recommended form: arm() right before the last pass
    e2 = red.early_buckets
    want = run(False, False)              # nothing leaves early
    e3 = red.early_buckets
    if rank == 0:
        scale = float(want.abs().max())
        print(json.dumps({"err_armed_first": float((got_first - want).abs().max()) / scale,
                          "err_armed_last": float((got_last - want).abs().max()) / scale,
                          "early": [e1 - e0, e2 - e1, e3 - e2], "gmax": scale}))
    dist.barrier()
    dist.destroy_process_group()


def _offset(red, p):
    return (red.views[p].data_ptr() - red.flat.data_ptr()) // 4


if __name__ == "__main__":
    main()
