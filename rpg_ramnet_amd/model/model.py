"""Drop-in for RAM_Net/model/model.py: ``ERGB2DepthRecurrent`` / ``ERGB2Depth`` with the reference's constructor
(config dict, same keys/defaults, model.py:12-77), call contract (model.py:141-219, :93-111), return structure and
``state_dict`` layout — computed by hand-written HIP kernels on MI355X.

    preds, super_states, states_lstm = model(item, prev_super_states, prev_states_lstm)

`item` holds NCHW float tensors (CPU or device) exactly as the reference's data loader yields them; predictions are
NCHW [B,1,H,W] in [0,1]; states are returned as NCHW-shaped views of the NHWC (channels_last) device buffers and
are accepted back unchanged (zero copy).  A previous state is never modified in place.
"""
import logging

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .statenet import StateNetPhasedRecurrent
from .unet import UNet


class BaseModel(nn.Module):
    """Same surface as RAM_Net/base/base_model.py:6-30 (config, logger, summary())."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.logger = logging.getLogger(self.__class__.__name__)

    def forward(self, *input):
        raise NotImplementedError

    def summary(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        self.logger.info('Trainable parameters: {}'.format(params))
        self.logger.info(self)


def _to_nchw(s):
    """NHWC buffer -> NCHW-shaped view (what callers of the reference expect to index/plot)."""
    if s is None:
        return None
    if isinstance(s, (list, tuple)):
        return type(s)(_to_nchw(t) for t in s)
    return s.permute(0, 3, 1, 2)


def _to_nhwc(s):
    if s is None:
        return None
    if isinstance(s, (list, tuple)):
        return type(s)(_to_nhwc(t) for t in s)
    return s.permute(0, 2, 3, 1)


def _state_nhwc(s, C):
    """One scale of a multi-scale state for the primitive API: an NHWC buffer (init_states / a previous update) is taken as it
    is, an NCHW-shaped view (what forward() returns) is viewed back; ConvLSTM (h, c) pairs element by element."""
    if isinstance(s, (list, tuple)):
        return type(s)(_state_nhwc(t, C) for t in s)
    if torch.is_tensor(s) and s.dim() == 4 and s.shape[-1] == C and s.is_contiguous():
        return s
    return _to_nhwc(s)


def _leaf(s):
    """The h tensor of a per-scale state (ConvLSTM states are (h, c) pairs)."""
    return s[0] if isinstance(s, (list, tuple)) else s


def _cat_batch(ts):
    """Concatenate along the batch axis; tensors that already lie one behind the other in one allocation (the K grids of a package out
    of the batched voxelizer) become a view instead of a copy."""
    t0 = ts[0]
    step = t0.numel() * t0.element_size()
    if all(t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and t.device == t0.device and not t.requires_grad
           and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() and t.data_ptr() == t0.data_ptr() + i * step
           for i, t in enumerate(ts)):
        return t0.as_strided((len(ts) * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(ts, 0)


class BaseERGB2Depth(BaseModel):
    def __init__(self, config):
        super().__init__(config)
        assert ('num_bins_rgb' in config)
        self.num_bins_rgb = int(config['num_bins_rgb'])
        assert ('num_bins_events' in config)
        self.num_bins_events = int(config['num_bins_events'])
        self.skip_type = str(config.get('skip_type', 'sum'))
        self.state_combination = str(config.get('state_combination', 'sum'))
        self.num_encoders = int(config.get('num_encoders', 4))
        self.base_num_channels = int(config.get('base_num_channels', 32))
        self.num_residual_blocks = int(config.get('num_residual_blocks', 2))
        self.recurrent_block_type = str(config.get('recurrent_block_type', 'convlstm'))
        self.norm = str(config['norm']) if 'norm' in config else None
        self.use_upsample_conv = bool(config.get('use_upsample_conv', True))
        self.every_x_rgb_frame = config.get('every_x_rgb_frame', 1)
        self.baseline = config.get('baseline', False)
        self.loss_composition = config.get('loss_composition', False)
        self.kernel_size = int(config.get('kernel_size', 5))
        self.gpu = torch.device('cuda:' + str(config['gpu']))     # KeyError when absent, like model.py:77
        self.full_frame = False
        self._crops = {}

    def set_full_frame(self, on=True):
        """Full-frame mode (not in the reference's model; its helper utils/inference_utils.py:287-314 CropParameters describes it): inputs
        whose height / width are not multiples of 2^num_encoders — the raw 260 x 346 frames, which the reference network itself cannot
        run — are reflect-padded to the next multiple (264 x 352) inside the input repack, the states live at the
        padded size, and every prediction is cropped back to the frame.  Off (default): such inputs fail as in the reference."""
        self.full_frame = bool(on)
        return self

    def _crop_for(self, H, W):
        """CropParameters of an H x W input in full-frame mode, or None when nothing has to be padded."""
        if not self.full_frame:
            return None
        key = (H, W)
        if key not in self._crops:
            self._crops[key] = ops.CropParameters(W, H, self.num_encoders)
        c = self._crops[key]
        return None if c.identity else c


class ERGB2Depth(BaseERGB2Depth):
    def __init__(self, config):
        super().__init__(config)
        self.unet = UNet(num_input_channels=self.num_bins_rgb, num_output_channels=1, skip_type=self.skip_type,
                         activation='sigmoid', num_encoders=self.num_encoders,
                         base_num_channels=self.base_num_channels, num_residual_blocks=self.num_residual_blocks,
                         norm=self.norm, use_upsample_conv=self.use_upsample_conv)

    def forward(self, item, prev_super_states, prev_states_lstm):
        crop = self._crop_for(*item["image"].shape[2:])
        x = ops.pack_input(item["image"], self.gpu, crop)
        pred = self.unet(x)
        return {"image": pred if crop is None else crop.crop(pred)}, {'image': None}, prev_states_lstm


class ERGB2DepthRecurrent(BaseERGB2Depth):
    """Recurrent asynchronous multimodal network: per-modality encoders update a shared multi-scale state."""

    def __init__(self, config):
        super().__init__(config)
        self.statenetphasedrecurrent = StateNetPhasedRecurrent(
            num_input_channels_rgb=self.num_bins_rgb, num_input_channels_events=self.num_bins_events,
            num_output_channels=1, skip_type=self.skip_type, state_combination=self.state_combination,
            activation='sigmoid', num_encoders=self.num_encoders, base_num_channels=self.base_num_channels,
            num_residual_blocks=self.num_residual_blocks, norm=self.norm, use_upsample_conv=self.use_upsample_conv,
            recurrent_block_type=self.recurrent_block_type, baseline=self.baseline)
        self.max_num_channels = self.base_num_channels * pow(2, self.num_encoders)

    # -- primitive API for irregular schedules (BASELINE.json config 4): one modality update / one decode per call
    def init_states(self, B, H, W):
        pair = (not bool(self.baseline)) and self.state_combination == 'convlstm'
        out = []
        for i in range(self.num_encoders):
            shp = [B, int(H / pow(2, i + 1)), int(W / pow(2, i + 1)), int(self.base_num_channels * pow(2, i + 1))]
            out.append([torch.zeros(shp, device=self.gpu), torch.zeros(shp, device=self.gpu)] if pair
                       else torch.zeros(shp, device=self.gpu))
        return out

    def update_events(self, events, states, lstm_state=None, out=None):
        """Asynchronous primitive (irregular schedules, BASELINE configs[3]): fold ONE event voxel grid [B,Ce,H,W] into the
        shared multi-scale state.  `states`: list returned by init_states()/a previous update (NHWC buffers or the
        NCHW-shaped views forward() returns).  Returns (new_states, lstm_state); nothing is modified in place.
        Equivalent to one iteration of the k-loop of model.py:176-195 without the decode.  out: optional per-scale NHWC buffers (a second
        init_states() set, never the one passed as `states`) that receive — and are returned as — the new state."""
        assert not bool(self.baseline), "baselines have no event branch (model.py:181-185)"
        st = [_state_nhwc(s, self.base_num_channels * 2 ** (i + 1)) for i, s in enumerate(states)]
        return self.statenetphasedrecurrent.forward_events(ops.pack_input(events, self.gpu, self._crop_for(*events.shape[2:])), st, lstm_state, out=out)

    def update_image(self, image, states, lstm_state=None, out=None):
        """Fold ONE frame [B,Cr,H,W] into the shared state (model.py:196-213 without the decode)."""
        st = [_state_nhwc(s, self.base_num_channels * 2 ** (i + 1)) for i, s in enumerate(states)]
        return self.statenetphasedrecurrent.forward_images(ops.pack_input(image, self.gpu, self._crop_for(*image.shape[2:])), st, lstm_state, out=out)

    def decode(self, states, frame_hw=None):
        """Depth prediction [B,1,H,W] in [0,1] from the current state (statenet.py:290-315).  frame_hw (full-frame mode): the (height,
        width) of the frames the state was built from — the prediction is cropped back to it."""
        pred = self.statenetphasedrecurrent.forward_decoder(states)
        crop = self._crop_for(*frame_hw) if frame_hw is not None else None
        return pred if crop is None else crop.crop(pred)

    def _forward_time_batched(self, item, states, crop, decode, emit):
        """One package (model.py:176-213) with everything that does not depend on the ORDER of the state updates batched over time
        (the training-step form of graph.TimeBatchedStream): RAM-Net's wiring (statenet.py:215-237) chains the ENCODER
        feature through the scales and only updates the shared state, and the decoder of measurement k reads the state after update k
        and nothing else.  So the K event grids go through head + encoders as ONE chain at batch K x B, the K + 1 state updates run one
        by one — each cell writes its new state into slot k of a [K + 1, B, h, w, C] buffer per scale — and the K + 1 decodes run as
        (at most) two chains over slots of those buffers: the measurements `loss_composition` supervises in one group, the others in
        another, so that the backward pass only walks the decodes that carry a gradient (every prediction keeps its autograd path: a
        caller that differentiates an unsupervised one is merely slower).  Per sample the arithmetic is what the pass-by-pass path does
        (the launches pick their tiling by grid size, so results agree to rounding, not bit for bit).  ConvGRU or ConvLSTM state,
        plain-conv encoders, no batch-statistics norm; anything else takes the pass-by-pass path."""
        net, K, n = self.statenetphasedrecurrent, self.every_x_rgb_frame, self.num_encoders
        keys = ['events{}'.format(k) for k in range(K)] + ['image']
        ev = _cat_batch([item[k].to(device=self.gpu, dtype=torch.float32) for k in keys[:K]])
        x = net.head_events(ops.pack_input(ev, self.gpu, crop))
        B = x.shape[0] // K
        feats = []
        for i, enc in enumerate(net.encoders_events):
            x = enc(x)
            pm = ops.premask_relu_feature(x)       # every gradient of x arrives through the fan-in below: it applies the encoder's ReLU mask
            if i + 1 < n:          # feeds the next encoder as a whole and the K cells of its scale slice by slice
                fan = ops.TimeFan.apply(x, K, pm)
                x, parts = fan[0], fan[1:]
            else:
                parts = ops.TimeSplit.apply(x, K, pm)
            feats.append(parts)
        # slot k of arena[i] = the state of scale i after update k (slot K: after the frame); written by the cells, never by a torch op
        pair = net.state_combination == 'convlstm'           # (h, c) per scale: the decoders read h
        # (ops.arena_slots: the slots are aliases with version counters of their own, not views of the buffer)
        if pair:
            arena = [[ops.arena_slots(K + 1, t.shape, self.gpu) for t in s] for s in states]
        else:
            arena = [ops.arena_slots(K + 1, s.shape, self.gpu) for s in states]
        slot = (lambda i, k: [a[1][k] for a in arena[i]]) if pair else (lambda i, k: arena[i][1][k])
        hbuf = (lambda i: arena[i][0][0]) if pair else (lambda i: arena[i][0])
        hof = (lambda s: s[0]) if pair else (lambda s: s)
        lc = self.loss_composition
        sup = [k for k in range(K + 1) if keys[k] in lc] if isinstance(lc, (list, tuple)) else None
        if not torch.is_grad_enabled():
            groups = [list(range(K + 1))]
        elif sup is None:
            groups = [[k] for k in range(K + 1)]              # unknown supervision: one decode per measurement, as in the reference
        else:
            groups = [g for g in ([k for k in range(K + 1) if k not in sup], sup) if g]
        md = ops.time_batch_max_decodes()
        if md > 0:
            groups = [g[j:j + md] for g in groups for j in range(0, len(g), md)]
        per_slot, preds = [], {}
        fuse = self._si_request(item, keys, crop)

        def decode_group(g):
            runs = [[g[0]]]                                    # maximal runs of consecutive slots (joined without a copy)
            for k in g[1:]:
                if k == runs[-1][-1] + 1:
                    runs[-1].append(k)
                else:
                    runs.append([k])
            joined = []
            for i in range(n):
                buf = hbuf(i)
                parts = [ops.TimeJoin.apply(buf[r[0]:r[-1] + 1].view((len(r) * B,) + tuple(buf.shape[2:])),
                                            *[hof(per_slot[k][i]) for k in r]) if len(r) > 1 else hof(per_slot[r[0]][i]) for r in runs]
                h = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
                joined.append((h,) if pair else h)
            order = [k for r in runs for k in r]
            if fuse is not None and all(keys[k] in fuse[2] for k in order):
                # every measurement of this chain is supervised and its target is at hand: the scale-invariant loss rides in the
                # prediction layer's launches (ops.PredSigmoidSI); trainer.sequence_loss picks the terms up from the predictions
                pred, losses = decode(joined, (fuse[0], fuse[1], [fuse[2][keys[k]] for k in order]))
                for j, k in enumerate(order):
                    preds[k] = pred[j * B:(j + 1) * B]
                    preds[k]._si_fused = (losses[j], fuse[2][keys[k]].data_ptr(), (fuse[0], fuse[1]))
                return
            pred = decode(joined)
            for j, k in enumerate(order):
                preds[k] = pred[j * B:(j + 1) * B]

        for k in range(K + 1):
            if k < K:
                states = [net.state_combination_events[i](feats[i][k], states[i], slot(i, k))[1] for i in range(n)]
            else:
                ximg = ops.pack_input(item['image'], self.gpu, crop)
                states, _ = net.forward_images(ximg, states, None, out=[slot(i, K) for i in range(n)])
            per_slot.append(states)
            for g in groups:                                   # a group is decoded as soon as its last update exists
                if g[-1] == k:
                    decode_group(g)
        for k in range(K + 1):
            emit(keys[k], preds[k], per_slot[k], {'encoders': [None] * n, 'state_comb': per_slot[k]})

    def _si_request(self, item, keys, crop):
        """(weight, n_lambda, {key: device target}) when this package's trainer asked for the scale-invariant loss of the supervised
        predictions to be formed inside the prediction layer (`self._si_fuse`, set around the call by trainer.sequence_loss) and it can be:
        gradients on, plain prediction layer, no full-frame crop, a [B, 1, H, W] target per requested key in the item."""
        req = getattr(self, "_si_fuse", None)
        if req is None or not torch.is_grad_enabled() or crop is not None or self.statenetphasedrecurrent.norm in ('BN', 'IN'):
            return None
        tg = {}
        shape = tuple(item['image'].shape[:1]) + (1,) + tuple(item['image'].shape[2:])
        for k in req["keys"]:
            t = item.get('depth_' + k) if k in keys else None
            if t is None:
                continue
            t = t.to(device=self.gpu, dtype=torch.float32)
            if tuple(t.shape) != shape:
                return None
            tg[k] = t.contiguous()
        return (float(req["weight"]), float(req["n_lambda"]), tg) if tg else None

    def forward(self, item, prev_super_states, prev_states_lstm):
        net = self.statenetphasedrecurrent
        predictions_dict, super_state_dict, states_lstm_dict = {}, {}, {}
        crop = self._crop_for(*item['image'].shape[2:])
        if prev_super_states is None:
            B, _, H, W = item['image'].shape
            states = self.init_states(B, H, W) if crop is None else self.init_states(B, crop.height_crop_size, crop.width_crop_size)
        else:
            states = [_to_nhwc(s) for s in prev_super_states]
        K, baseline, lc = self.every_x_rgb_frame, self.baseline, self.loss_composition

        def lstm_in(d):       # states_lstm dicts cross the boundary NCHW-shaped too
            return None if d is None else {'encoders': [_to_nhwc(s) for s in d['encoders']],
                                           'state_comb': [_to_nhwc(s) for s in d['state_comb']]}

        def emit(key, pred, ss, sl):
            predictions_dict[key] = pred if crop is None else crop.crop(pred)
            views = [_to_nchw(s) for s in ss]
            super_state_dict[key] = views
            comb = []
            for s_in, v, c in zip(ss, views, sl['state_comb']):
                comb.append(v if c is s_in else _to_nchw(c))         # GRU mode: state_comb[i] IS super_state[i]
            states_lstm_dict[key] = {'encoders': [_to_nchw(s) for s in sl['encoders']], 'state_comb': comb}

        side = ops.decode_stream(self.gpu) if ops.decoder_overlap() else None

        def decode(ss, si=None):
            """Prediction from the state after this update; optionally on the decode stream, concurrent with the next update.
            si: see StateNetPhasedRecurrent.forward_decoder — returns (prediction, losses) then."""
            if side is None:
                return net.forward_decoder(ss, si)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                pred = net.forward_decoder(ss, si)
            if not torch.cuda.is_current_stream_capturing():       # (inside a capture the graph's edges order the private pool)
                for t in ss:
                    for u in (t if isinstance(t, (list, tuple)) else (t,)):
                        u.record_stream(side)
            return pred

        events_as_image = baseline == "ergb0" or (baseline == "e" and lc == "image")
        last = None
        if (ops.time_batching() and not bool(baseline) and K >= 2 and net.recurrent_block_type == 'conv'
                and net.state_combination in ('convgru', 'convlstm') and net.norm not in ('BN', 'IN')
                and all(item['events{}'.format(k)].shape == item['events0'].shape for k in range(K))
                # (the kernels address a tensor with 32-bit element offsets: the K x B head output must stay below 2^32 elements)
                and (K + 1) * item['events0'].shape[0] * 4 * _leaf(states[0]).shape[1] * _leaf(states[0]).shape[2] * self.base_num_channels < (1 << 32)):
            self._forward_time_batched(item, states, crop, decode, emit)
            if side is not None:                       # predictions are consumed on the caller's stream
                torch.cuda.current_stream().wait_stream(side)
                if not torch.cuda.is_current_stream_capturing():
                    for pred in predictions_dict.values():
                        pred.record_stream(torch.cuda.current_stream())
                        fused = getattr(pred, "_si_fused", None)       # (the fused loss term was allocated on the decode stream too)
                        if fused is not None and torch.is_tensor(fused[0]):
                            fused[0].record_stream(torch.cuda.current_stream())
            return predictions_dict, super_state_dict, states_lstm_dict
        if not bool(baseline) or events_as_image:
            if events_as_image:
                loop_range, last = K - 1, lstm_in(prev_states_lstm['image'])
            else:
                loop_range, last = K, lstm_in(prev_states_lstm['events{}'.format(K - 1)])
            for k in range(loop_range):
                x = ops.pack_input(item['events{}'.format(k)], self.gpu, crop)
                if baseline == "ergb0" or baseline == 'e':
                    ss, sl = net.forward_images(x, states, last)
                else:
                    ss, sl = net.forward_events(x, states, last)
                emit('events{}'.format(k), decode(ss), ss, sl)
                states, last = ss, sl

        x = ops.pack_input(item['image'], self.gpu, crop)
        if not bool(baseline) or baseline == "rgb" or (baseline == "e" and lc != "image"):
            last = lstm_in(prev_states_lstm['image'])
        ss, sl = net.forward_images(x, states, last)
        emit('image', decode(ss), ss, sl)
        if side is not None:                       # predictions are consumed on the caller's stream
            torch.cuda.current_stream().wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for pred in predictions_dict.values():
                    pred.record_stream(torch.cuda.current_stream())
        return predictions_dict, super_state_dict, states_lstm_dict
