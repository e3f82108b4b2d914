#!/usr/bin/env python3
"""Golden fixture for the epoch-level trainer (rpg_ramnet_amd.trainer.EpochTrainer), produced by RUNNING the reference's
BaseTrainer (RAM_Net/base/base_trainer.py) with a scripted `_train_epoch`: the learning-rate sequence, the monitor_best sequence,
the checkpoint files left in the directory and the keys / scalar fields of a checkpoint.  Run in the build container only:
    python tests/golden/make_golden_trainer.py
"""
import json
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

LOSSES = [0.9, 0.7, 0.8, 0.6, 0.65, 0.5, 0.55]          # scripted 'loss' per epoch; 'val_loss' = loss + 0.1 * (-1)^epoch
CONFIG = {"name": "golden", "cuda": True, "gpu": 0, "optimizer_type": "Adam", "optimizer": {"lr": 3e-4, "weight_decay": 0, "amsgrad": True},
          "lr_scheduler_type": "ExponentialLR", "lr_scheduler_freq": 2, "lr_scheduler": {"gamma": 0.5},
          "trainer": {"epochs": 7, "save_freq": 3, "verbosity": 0, "monitor": "val_loss", "monitor_mode": "min"}}


def scripted(epoch):
    loss = LOSSES[epoch - 1]
    return {"loss": loss, "val_loss": loss + 0.1 * (-1) ** epoch}


def main():
    import torch._dynamo  # noqa: F401  (optimizer constructors import it lazily; it must not meet the stub modules below)
    import_reference()
    import base.base_trainer as bt
    from logger import Logger

    class Writer:                     # the stubbed SummaryWriter: swallow everything
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    bt.SummaryWriter = Writer

    class T(bt.BaseTrainer):
        def _train_epoch(self, epoch):
            self.mb.append(self.monitor_best)
            return scripted(epoch)

    torch.manual_seed(0)
    net = torch.nn.Conv2d(1, 2, 3)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = json.loads(json.dumps(CONFIG))
        cfg["trainer"]["save_dir"] = tmp
        lrs = []
        t = T(net, None, None, [], None, cfg, train_logger=Logger())
        t.mb = []
        # learning rate after every epoch: wrap the scheduler's step bookkeeping by reading it at the next epoch start
        orig = t._train_epoch

        def wrapped(epoch):
            lrs.append(t.optimizer.param_groups[0]["lr"])
            return orig(epoch)
        t._train_epoch = wrapped
        t.train()
        lrs.append(t.optimizer.param_groups[0]["lr"])
        files = sorted(os.listdir(os.path.join(tmp, "golden")))
        ck = torch.load(os.path.join(tmp, "golden", "model_best.pth.tar"), weights_only=False)
        out = {"config": CONFIG, "losses": LOSSES, "lr_at_epoch_start": lrs[:-1], "lr_after_last_epoch": lrs[-1],
               "monitor_best_at_epoch_start": [m if m != float("inf") else "inf" for m in t.mb], "monitor_best_final": t.monitor_best,
               "files": [f for f in files if f.endswith(".pth.tar")], "checkpoint_keys": sorted(ck.keys()),
               "best": {"epoch": ck["epoch"], "monitor_best": ck["monitor_best"], "arch": ck["arch"],
                        "logger_entries": len(ck["logger"].entries), "optimizer_keys": sorted(ck["optimizer"].keys())}}
    with open(os.path.join(HERE, "trainer.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
