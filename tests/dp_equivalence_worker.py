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
    red = FlatGradReducer(model)
    rng = np.random.default_rng(21)
    n_seq, L = 4, 2                                                      # 4 sequences of 2 packages, sharded r, r + world, ...
    data = [[make_item(rng, 1, 32, 48, 2, 5, 1, True, 0.1) for _ in range(L)] for _ in range(n_seq)]

    def batch(idx):
        return [{k: torch.cat([data[i][l][k] for i in idx]) for k in data[0][l]} for l in range(L)]

    def local_grads(idx):
        red.zero()
        total, _ = sequence_loss(model, batch(idx), cfg["loss_composition"], [1, 1])
        total.backward()
        torch.cuda.synchronize()
        return red.flat.clone()

    mine = shard_indices(n_seq, rank, world)
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


if __name__ == "__main__":
    main()
