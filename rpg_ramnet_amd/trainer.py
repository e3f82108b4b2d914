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


def sequence_loss(model, sequence, loss_composition, loss_weights, loss_params=None, grad_loss_weight=None):
    """BPTT over the L packages of `sequence` (list of item dicts with 'depth_<key>' targets).
    grad_loss_weight: weight of the multi-scale gradient loss (config['grad_loss']['weight'], 0.25 in the released
    recipe; lstm_trainer.py:162-168, :197-199) or None for the SI loss alone.
    Returns (loss to call .backward() on, loss value the reference would report)."""
    loss_params = loss_params or {"weight": 1.0, "n_lambda": 1.0}
    gterms = []
    L = len(sequence)
    assert L > 0
    K = model.every_x_rgb_frame
    prev_super, prev_lstm = None, empty_states_lstm(K)
    terms, keys_seen = [], []
    for item in sequence:
        preds, supers, lstms = model(item, prev_super, prev_lstm)
        for key, value in preds.items():
            if not loss_composition or key in loss_composition:
                w = loss_weights[loss_composition.index(key)]
                target = item['depth_' + key].to(model.gpu)
                terms.append(w * ops.scale_invariant_loss(value, target, **loss_params))
                if grad_loss_weight is not None:
                    gterms.append(w * ops.multi_scale_grad_loss(value, target))
                if key not in keys_seen:
                    keys_seen.append(key)
        prev_super, prev_lstm = supers['image'], lstms
    total = torch.stack(terms).sum() / float(L)
    if grad_loss_weight is not None:
        total = total + grad_loss_weight * torch.stack(gterms).sum() / float(L)
    return total, total.detach() * len(keys_seen)
