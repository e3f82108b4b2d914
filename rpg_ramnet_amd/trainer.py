"""Sequence-level driver: the call pattern and loss assembly of the reference's
LSTMTrainer.forward_pass_sequence (RAM_Net/trainer/lstm_trainer.py:228-390, :152-226) around the HIP model.

Quirk mirrored: the reference aliases ONE loss dict for every supervised key and adds the second key's
total under no_grad, so the DIFFERENTIATED scalar is (sum_{l,key} w_key SI)/L while the REPORTED loss is that value
times the number of supervised keys.  `sequence_loss` returns both.
"""
import torch

from . import ops


def empty_states_lstm(K):
    d = {}
    for k in range(K):
        d['events{}'.format(k)] = None
        d['depth{}'.format(k)] = None
    d['image'] = None
    return d


LOSS_SEMANTICS = {
    False: "per-rank mean over the rank's batch, gradients averaged over ranks (standard DDP)",
    True: "exact global batch: SI statistics (sum d, sum d^2, n per supervised map) all-reduced before the backward; the N-rank "
          "gradient equals the single-rank gradient on the concatenated batch (model/loss.py:9)",
}


LOSS_TYPES = ("scale_invariant_loss", "scale_invariant_log_loss", "mse_loss")      # config['loss']['type'] (train.py:192 getattr(module_loss, ...))


def _nominal_loss(loss_type, value, target, loss_params):
    if loss_type == "scale_invariant_loss":
        return ops.scale_invariant_loss(value, target, **loss_params)
    if loss_type == "scale_invariant_log_loss":
        return ops.scale_invariant_log_loss(value, target, **{k: v for k, v in loss_params.items() if k == "n_lambda"})
    if loss_type == "mse_loss":
        return ops.mse_loss(value, target)
    raise KeyError(loss_type)


def sequence_loss(model, sequence, loss_composition, loss_weights, loss_params=None, grad_loss_weight=None, dp_exact=False,
                  process_group=None, loss_type="scale_invariant_loss", mse_loss=None):
    """BPTT over the L packages of `sequence` (list of item dicts with 'depth_<key>' targets).
    grad_loss_weight: weight of the multi-scale gradient loss (config['grad_loss']['weight'], 0.25 in the released
    recipe; lstm_trainer.py:162-168, :197-199) or None for the SI loss alone.
    dp_exact (data parallel; default off): the reference's loss takes mean(d)^2 over the WHOLE batch (model/loss.py:9),
    which an average of per-rank losses is not.  With dp_exact every rank computes (sum d, sum d^2, n) of each of its supervised
    maps, ONE all-reduce sums the [terms x 4] table over the ranks before the backward, and every term's value and gradient follow
    from the global sums (ops.SILossFromStats) — after the reducer's gradient average the N-rank gradient is the single-rank one on
    the concatenated batch, and every rank reports the global loss.  (The multi-scale gradient loss normalises per scale by its own
    valid-pixel count: it stays per rank.)
    loss_type: config['loss']['type'] — 'scale_invariant_loss' (every shipped config), 'scale_invariant_log_loss' (model/loss.py:12-15)
    or 'mse_loss' (:18-19); loss_params = config['loss']['config'].
    mse_loss: config['mse_loss'] (lstm_trainer.py:76-90) — {'weight': 1.0, 'downsampling_factor': 0.5} — adds
    weight * sum_terms w_key * mse(bilinear x factor of prediction and target) / L (lstm_trainer.py:169-185, :205-208), or None.
    Returns (loss to call .backward() on, loss value the reference would report)."""
    if loss_params is None:
        loss_params = {"weight": 1.0, "n_lambda": 1.0} if loss_type == "scale_invariant_loss" else {}
    if dp_exact:
        assert loss_type == "scale_invariant_loss" and mse_loss is None, "dp_exact: the scale-invariant loss only"
        return _sequence_loss_dp_exact(model, sequence, loss_composition, loss_weights, loss_params, grad_loss_weight, process_group)
    gterms, mterms = [], []
    L = len(sequence)
    assert L > 0
    K = model.every_x_rgb_frame
    prev_super, prev_lstm = None, empty_states_lstm(K)
    terms, keys_seen = [], []
    # the scale-invariant loss of the supervised predictions inside the prediction layer's own launches (ops.PredSigmoidSI): the model is
    # told which keys and parameters for the duration of its forward call and hangs the loss term on the prediction it returns
    si_par = (float(loss_params.get("weight", 1.0)), float(loss_params.get("n_lambda", 1.0))) if loss_type == "scale_invariant_loss" else None
    ask = si_par is not None and ops.si_fusion() and isinstance(loss_composition, (list, tuple)) and hasattr(model, "_si_request")
    for item in sequence:
        if ask:
            model._si_fuse = {"weight": si_par[0], "n_lambda": si_par[1], "keys": list(loss_composition)}
            # the supervised targets move to the device ONCE (a DataLoader item holds CPU tensors): the model's fused loss and the term below
            # then see the same tensor — a second copy would make the pointers differ, the fused term be dropped and its launches wasted
            moved = {}
            for key in loss_composition:
                t = item.get('depth_' + key)
                if torch.is_tensor(t) and (t.device != torch.device(model.gpu) or t.dtype != torch.float32 or not t.is_contiguous()):
                    moved['depth_' + key] = t.to(device=model.gpu, dtype=torch.float32).contiguous()
            if moved:
                item = dict(item, **moved)
        try:
            preds, supers, lstms = model(item, prev_super, prev_lstm)
        finally:
            if ask:
                model._si_fuse = None
        for key, value in preds.items():
            if not loss_composition or key in loss_composition:
                w = loss_weights[loss_composition.index(key)]
                target = item['depth_' + key].to(model.gpu)
                fused = getattr(value, "_si_fused", None)
                if fused is not None and ask and fused[2] == si_par and target.dtype == torch.float32 and fused[1] == target.data_ptr():
                    terms.append(w * fused[0])
                else:
                    terms.append(w * _nominal_loss(loss_type, value, target, loss_params))
                if grad_loss_weight is not None:
                    gterms.append(w * ops.multi_scale_grad_loss(value, target))
                if mse_loss is not None:
                    mterms.append(w * ops.mse_loss(value, target, mse_loss.get('downsampling_factor', 0.5)))
                if key not in keys_seen:
                    keys_seen.append(key)
        prev_super, prev_lstm = supers['image'], lstms
    total = torch.stack(terms).sum() / float(L)
    if grad_loss_weight is not None:
        total = total + grad_loss_weight * torch.stack(gterms).sum() / float(L)
    if mse_loss is not None:
        total = total + float(mse_loss.get('weight', 1.0)) * torch.stack(mterms).sum() / float(L)
    return total, total.detach() * len(keys_seen)


def _sequence_loss_dp_exact(model, sequence, loss_composition, loss_weights, loss_params, grad_loss_weight, group):
    import torch.distributed as dist
    L = len(sequence)
    assert L > 0
    K = model.every_x_rgb_frame
    prev_super, prev_lstm = None, empty_states_lstm(K)
    sup, gterms, keys_seen = [], [], []
    for item in sequence:
        preds, supers, lstms = model(item, prev_super, prev_lstm)
        for key, value in preds.items():
            if not loss_composition or key in loss_composition:
                w = loss_weights[loss_composition.index(key)]
                target = item['depth_' + key].to(model.gpu).float()
                sup.append((w, value.float(), target))
                if grad_loss_weight is not None:
                    gterms.append(w * ops.multi_scale_grad_loss(value, target))
                if key not in keys_seen:
                    keys_seen.append(key)
        prev_super, prev_lstm = supers['image'], lstms
    table = torch.empty(len(sup), 4, device=model.gpu, dtype=torch.float64)
    for i, (_, value, target) in enumerate(sup):
        ops.si_local_stats(value.detach(), target, table[i])
    world = 1
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)       # stream-ordered on the compute stream: 32 B per term
    terms = [w * ops.SILossFromStats.apply(value, target, table[i], float(loss_params["weight"]), float(loss_params["n_lambda"]),
                                           float(world)) for i, (w, value, target) in enumerate(sup)]
    total = torch.stack(terms).sum() / float(L)
    if grad_loss_weight is not None:
        total = total + grad_loss_weight * torch.stack(gterms).sum() / float(L)
    # value: SILossFromStats.forward returns the global loss itself (the gain only scales its backward)
    return total, total.detach() * len(keys_seen)


class EpochTrainer:
    """Epoch-level half of the reference trainer (RAM_Net/base/base_trainer.py:16-61 constructor, :65-123 `train`, :133-158
    `_save_checkpoint`, :160-179 `_resume_checkpoint`) around a user-supplied epoch function — NOT the control plane: no
    TensorBoard, no previews, no metric plumbing.

    config keys used exactly as the reference reads them: 'name', 'optimizer_type' + 'optimizer', 'lr_scheduler_type' +
    'lr_scheduler' + 'lr_scheduler_freq', config['trainer']: 'epochs', 'save_freq', 'save_dir', 'monitor', 'monitor_mode'.
    ``train_epoch(epoch) -> dict`` returns the epoch's log entries (at least 'loss' and the monitored key).  Order of events per
    epoch, as in base_trainer.py:103-122: log entry -> best-checkpoint (saved under the epoch name, then renamed to
    'model_best.pth.tar') -> periodic checkpoint when epoch % save_freq == 0 -> scheduler step when epoch % lr_scheduler_freq == 0.
    Checkpoints have the reference's dict layout (checkpoint.save_checkpoint)."""

    def __init__(self, model, config, train_epoch, resume=None, train_logger=None, reducer=None):
        import math
        import os
        from . import checkpoint as ck
        self.model, self.config, self.train_epoch, self.reducer = model, config, train_epoch, reducer
        self.epochs = config['trainer']['epochs']
        self.save_freq = config['trainer']['save_freq']
        self.optimizer = getattr(torch.optim, config['optimizer_type'])(model.parameters(), **config['optimizer'])
        sched = getattr(torch.optim.lr_scheduler, config.get('lr_scheduler_type', ''), None)
        self.lr_scheduler = sched(self.optimizer, **config['lr_scheduler']) if sched else None
        self.lr_scheduler_freq = config.get('lr_scheduler_freq', 1)
        self.monitor, self.monitor_mode = config['trainer']['monitor'], config['trainer']['monitor_mode']
        assert self.monitor_mode in ('min', 'max')
        self.monitor_best = math.inf if self.monitor_mode == 'min' else -math.inf
        self.start_epoch = 1
        self.checkpoint_dir = os.path.join(config['trainer']['save_dir'], config['name'])
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        import json                      # base_trainer.py:50-51: the reference's test / evaluation tooling reads it beside the checkpoints
        with open(os.path.join(self.checkpoint_dir, 'config.json'), 'w') as f:
            json.dump(config, f, indent=4, sort_keys=False)
        self.train_logger = train_logger if train_logger is not None else ck.Logger()
        self.lr_history = []
        if resume:
            self.start_epoch, c = ck.resume(resume, model, self.optimizer, map_location='cpu')
            self.monitor_best = c['monitor_best']
            self.train_logger = c['logger']

    def _save(self, epoch, log, save_best=False):
        import os
        from . import checkpoint as ck
        path = ck.save_checkpoint(ck.checkpoint_name(self.checkpoint_dir, epoch, log['loss']), self.model, self.optimizer, epoch,
                                  self.config, self.monitor_best, self.train_logger)
        if save_best:
            best = os.path.join(self.checkpoint_dir, 'model_best.pth.tar')
            os.rename(path, best)
            path = best
        return path

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self.train_epoch(epoch)
            log = {'epoch': epoch}
            log.update({k: v for k, v in result.items() if 'previews' not in k})
            self.train_logger.add_entry(log)
            if (self.monitor_mode == 'min' and log[self.monitor] < self.monitor_best) or \
                    (self.monitor_mode == 'max' and log[self.monitor] > self.monitor_best):
                self.monitor_best = log[self.monitor]
                self._save(epoch, log, save_best=True)
            if epoch % self.save_freq == 0:
                self._save(epoch, log)
            if self.lr_scheduler and epoch % self.lr_scheduler_freq == 0:
                self.lr_scheduler.step()
            self.lr_history.append(self.lr_scheduler.get_last_lr()[0] if self.lr_scheduler else self.optimizer.param_groups[0]['lr'])
        return self.train_logger
