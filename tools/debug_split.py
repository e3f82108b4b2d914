import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpg_ramnet_amd import ops, _hip as H

dev = torch.device("cuda:0")
torch.manual_seed(1)
B, Hh, W, cin, cout, k = 2, 16, 32, 8, 32, 5
w = torch.nn.Parameter((torch.rand(cout, cin, k, k, device=dev) - 0.5) * 0.14)
b = torch.nn.Parameter(torch.zeros(cout, device=dev))
dy = torch.randn(B, Hh, W, cout, device=dev)
y = torch.randn(B, Hh, W, cout, device=dev)
res = {}
for prec in ["f32", "bf16x3"]:
    ops.set_precision(prec)
    cp = ops.ConvParam([w], [b])
    for mode in ["plain", "mask"]:
        dx = torch.zeros(B, Hh, W, cin, device=dev)
        if mode == "plain":
            ops.conv_launch(dy, ops.Taps.get("dgrad1", k, 2), cp.bwd(), dx, cin)
        else:
            ops.conv_launch(dy, ops.Taps.get("dgrad1", k, 2), cp.bwd(), dx, cin, xm=y, in_mode=H.IN_RELUMASK)
        torch.cuda.synchronize()
        res[(prec, mode)] = dx.cpu().numpy()
for mode in ["plain", "mask"]:
    a, r = res[("bf16x3", mode)], res[("f32", mode)]
    e = np.abs(a - r)
    print(mode, "max err", e.max(), "max ref", np.abs(r).max(), "n>1e-3:", (e > 1e-3).sum(), "of", e.size)
    idx = np.argwhere(e > 1e-3)
    if len(idx):
        print(" bad b:", np.unique(idx[:, 0]), "y:", np.unique(idx[:, 1]), "x:", np.unique(idx[:, 2]), "c:", np.unique(idx[:, 3]))
