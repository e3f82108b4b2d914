"""Checkpoint files in the reference's layout (RAM_Net/base/base_trainer.py:133-179):

    {'arch', 'epoch', 'logger', 'state_dict', 'optimizer', 'monitor_best', 'config'}   ->  *.pth.tar

so that checkpoints written here load in the reference's `test.py` / `--resume`, and the released checkpoints
(`ramnet_sim.pth.tar`, README.md:59) load here.  Reference files pickle a `logger.logger.Logger` instance under
'logger'; that module is not importable outside the reference tree, so loading installs a structural stand-in
(same attribute: `entries`) for the duration of `torch.load` — and `save_checkpoint` pickles its own logger under that
same module path (`logger.logger.Logger`), so a file written here unpickles inside the reference tree with the reference's
own class and WITHOUT `rpg_ramnet_amd` on `sys.path`.
"""
import os
import sys
import types

import torch


class Logger:
    """Same surface as RAM_Net/logger/logger.py:4-18 (dict of entries)."""

    def __init__(self):
        self.entries = {}

    def add_entry(self, entry):
        self.entries[len(self.entries) + 1] = entry

    def __str__(self):
        import json
        return json.dumps(self.entries, sort_keys=True, indent=4)


Logger.__module__ = 'logger.logger'     # pickled under the reference's module path (RAM_Net/logger/logger.py)


class _LoggerModules:
    """Make `logger.logger.Logger` resolve to THIS module's class while pickling / unpickling — also when another `logger`
    package (the reference tree's own, in a side-by-side setup) is already imported: its entries in `sys.modules` are saved,
    overridden for the duration of the call and restored afterwards (pickle checks that the class found under the pickled
    module path is the very object being saved)."""

    _NAMES = ('logger', 'logger.logger')

    def __enter__(self):
        self.saved = {m: sys.modules.get(m) for m in self._NAMES}
        pkg = types.ModuleType('logger')
        sub = types.ModuleType('logger.logger')
        sub.Logger = Logger
        pkg.Logger = Logger
        pkg.logger = sub
        sys.modules['logger'], sys.modules['logger.logger'] = pkg, sub
        return self

    def __exit__(self, *exc):
        for m, old in self.saved.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old


def _as_own_logger(logger):
    """A foreign `logger.logger.Logger` instance (the reference's class, same surface) re-wrapped into this module's class so
    that it pickles under the stand-in modules."""
    if logger is None:
        return Logger()
    if isinstance(logger, Logger):
        return logger
    own = Logger()
    own.entries = dict(getattr(logger, 'entries', {}))
    return own


def checkpoint_name(save_dir, epoch, loss):
    """base_trainer.py:151-152 naming."""
    return os.path.join(save_dir, 'checkpoint-epoch{:03d}-loss-{:.4f}.pth.tar'.format(epoch, loss))


def save_checkpoint(path, model, optimizer, epoch, config, monitor_best=float('inf'), logger=None):
    state = {
        'arch': type(model).__name__,
        'epoch': epoch,
        'logger': _as_own_logger(logger),
        'state_dict': model.state_dict(),
        'optimizer': optimizer.state_dict() if optimizer is not None else None,
        'monitor_best': monitor_best,
        'config': config,
    }
    with _LoggerModules():
        torch.save(state, path)
    return path


def load_checkpoint(path, map_location='cpu'):
    """torch.load of a reference-layout checkpoint (needs weights_only=False: the file pickles Python objects)."""
    with _LoggerModules():
        return torch.load(path, map_location=map_location, weights_only=False)


def resume(path, model, optimizer=None, map_location='cpu'):
    """base_trainer.py:160-179: restore model (strict), optimizer, epoch, monitor_best.  Returns (start_epoch, ckpt)."""
    ckpt = load_checkpoint(path, map_location)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer is not None and ckpt.get('optimizer') is not None:
        optimizer.load_state_dict(ckpt['optimizer'])
        dev = next(model.parameters()).device
        for st in optimizer.state.values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    st[k] = v.to(dev)
    return ckpt['epoch'] + 1, ckpt
