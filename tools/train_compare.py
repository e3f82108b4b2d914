#!/usr/bin/env python3
"""HIP path vs CPU oracle through several Adam steps on the same data: the loss trajectories must coincide."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from recipe import make_item  # noqa: E402
from oracle import ramnet_ref  # noqa: E402
from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent  # noqa: E402
from rpg_ramnet_amd.trainer import sequence_loss  # noqa: E402

K = 2
cfg = dict(num_bins_rgb=1, num_bins_events=5, skip_type="sum", recurrent_block_type="conv", state_combination="convgru",
           num_encoders=3, base_num_channels=32, num_residual_blocks=2, use_upsample_conv=True, norm="none", gpu=0,
           every_x_rgb_frame=K, baseline=False, loss_composition=["image", "events1"])
torch.manual_seed(0)
model = ERGB2DepthRecurrent(cfg)
sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
model = model.to(model.gpu).train()
rng = np.random.default_rng(0)
seq = [make_item(rng, 2, 32, 48, K, 5, 1, True, 0.0) for _ in range(2)]
for item in seq:
    tgt = 0.25 + 0.5 * torch.nn.functional.avg_pool2d(item["image"], 5, 1, 2)
    item["depth_image"], item["depth_events1"] = tgt, tgt.clone()
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
opt_g = torch.optim.Adam(model.parameters(), lr=lr)
opt_c = torch.optim.Adam(list(sd.values()), lr=lr)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    opt_g.zero_grad()
    lg, _ = sequence_loss(model, seq, cfg["loss_composition"], [1, 1])
    lg.backward()
    opt_g.step()
    opt_c.zero_grad()
    lc, _ = ramnet_ref.sequence_loss(sd, cfg, seq, cfg["loss_composition"], [1, 1])
    lc.backward()
    opt_c.step()
    wdiff = max(float((p.detach().cpu() - sd[k].detach()).abs().max()) for k, p in model.named_parameters())
    print("step %2d  hip %.6f  oracle %.6f  max|w_hip - w_oracle| %.2e" % (it, float(lg.detach()), float(lc.detach()), wdiff))
