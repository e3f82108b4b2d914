"""StateNetPhasedRecurrent on HIP kernels — same sub-module names, construction order (=> identical seeded
initialisation) and forward contract as RAM_Net/model/statenet.py:120-315.  Activations are NHWC internally.

Combinations that raise inside the reference itself are refused at construction (DESIGN.md "out of scope"):
state_combination 'sum'/'conv' (tuple-unpack of a tensor, statenet.py:230-233), skip_type != 'sum' for the
recurrent net (decoder 0 gets no skip, statenet.py:107 vs :302).
"""
import torch
import torch.nn as nn

from .. import ops
from .submodules import ConvLayer, UpsampleConvLayer, TransposedConvLayer, RecurrentConvLayer, Recurrent2ConvLayer, \
    ResidualBlock


class StateNetPhasedRecurrent(nn.Module):
    def __init__(self, num_input_channels_rgb, num_input_channels_events, num_output_channels=1, skip_type='sum',
                 state_combination='sum', activation='sigmoid', num_encoders=4, base_num_channels=32,
                 num_residual_blocks=2, norm=None, use_upsample_conv=True, recurrent_block_type='convlstm',
                 baseline=False):
        super().__init__()
        if skip_type not in ('sum', 'concat', 'no_skip', None):
            raise KeyError('Could not identify skip_type, please add "skip_type":'
                           ' "sum", "concat" or "no_skip" to config["model"]')
        if state_combination not in ('sum', 'conv', 'convlstm', 'convgru'):
            raise KeyError('Could not identify state_combination, please add "state_combination":'
                           ' "sum", "conv", "convlstm" or "convgru" to config["model"]')
        if skip_type != 'sum':
            raise NotImplementedError("recurrent net: only skip_type 'sum' runs in the reference (statenet.py:107 vs :302)")
        if state_combination not in ('convlstm', 'convgru'):
            raise NotImplementedError("state_combination %r raises in the reference (statenet.py:230-233)" % state_combination)
        assert activation == 'sigmoid' and num_output_channels == 1
        assert recurrent_block_type in ('conv', 'convlstm')
        self.skip_type, self.state_combination = skip_type, state_combination
        self.recurrent_block_type, self.norm, self.baseline = recurrent_block_type, norm, baseline
        self.num_encoders, self.base_num_channels = num_encoders, base_num_channels
        self.num_residual_blocks = num_residual_blocks
        self.max_num_channels = base_num_channels * pow(2, num_encoders)
        if use_upsample_conv:
            print('Using UpsampleConvLayer (slow, but no checkerboard artefacts)')
            Up = UpsampleConvLayer
        else:
            print('Using TransposedConvLayer (fast, with checkerboard artefacts)')
            Up = TransposedConvLayer
        enc_in = [base_num_channels * pow(2, i) for i in range(num_encoders)]
        enc_out = [base_num_channels * pow(2, i + 1) for i in range(num_encoders)]

        # --- construction order mirrors statenet.py:139-202 (RNG stream => same seeded weights)
        self.head_rgb = ConvLayer(num_input_channels_rgb, base_num_channels, kernel_size=5, stride=1, padding=2)
        self.encoders_rgb = nn.ModuleList()
        if not bool(baseline):
            self.head_events = ConvLayer(num_input_channels_events, base_num_channels, kernel_size=5, stride=1, padding=2)
            self.encoders_events = nn.ModuleList()
            self.state_combination_events = nn.ModuleList()
        self.state_combination_images = nn.ModuleList()
        for cin, cout in zip(enc_in, enc_out):
            if recurrent_block_type == 'convlstm':
                mk = lambda: Recurrent2ConvLayer(cin, cout, kernel_size=5, stride=2, padding=2, norm=norm,  # noqa: E731
                                                 recurrent_block_type='convlstm')
            else:
                mk = lambda: ConvLayer(cin, cout, kernel_size=5, stride=2, padding=2, norm=norm)  # noqa: E731
            self.encoders_rgb.append(mk())
            if not bool(baseline):
                self.encoders_events.append(mk())
                self.state_combination_events.append(RecurrentConvLayer(cout, cout, recurrent_block_type=state_combination))
            self.state_combination_images.append(RecurrentConvLayer(cout, cout, recurrent_block_type=state_combination))
        self.resblocks = nn.ModuleList([ResidualBlock(self.max_num_channels, self.max_num_channels, norm=norm)
                                        for _ in range(num_residual_blocks)])
        self.decoders = nn.ModuleList([Up(c, c // 2, kernel_size=5, padding=2, norm=norm) for c in reversed(enc_out)])
        self.pred = ConvLayer(base_num_channels, num_output_channels, 1, activation=None, norm=norm)

    # ---------------------------------------------------------------------------------------------- encoders
    def _encode(self, x, head, encoders, combs, prev_super_state, prev_states_lstm, feed_state_forward, out=None):
        """out (RAM-Net state updates only): per-scale NHWC buffers — (h, c) pairs for convlstm — that receive the new state (the
        streaming runtimes' static buffers; none of them may alias the state being read)."""
        n = self.num_encoders
        assert out is None or not feed_state_forward
        x = head(x)
        if prev_states_lstm is None:
            prev_states_lstm = {'encoders': [None] * n, 'state_comb': [None] * n}
        super_states, states_lstm = [], {'encoders': [], 'state_comb': []}
        # inference: the state updates of the scales are mutually independent -> all but the last on streams of their own
        branch = (not feed_state_forward) and ops.branch_overlap() and not torch.is_grad_enabled() and x.is_cuda
        joins = []
        for i, encoder in enumerate(encoders):
            if self.recurrent_block_type == 'conv':
                x, enc_state = encoder(x), None
            else:
                x, enc_state = encoder(x, prev_states_lstm['encoders'][i])
            xc = x            # what the cell of this scale reads
            if not feed_state_forward and enc_state is None and torch.is_grad_enabled() and ops.premask_relu_feature(x):
                # training: the fan-in of x's gradient (cell + next encoder; autograd's own add otherwise) applies the encoder's ReLU mask,
                # so that the encoder's backward launches read ONE plain operand (ops.set_relu_premask; as in the time-batched path)
                if i + 1 < n:
                    x, xc = ops.TimeFan.apply(x, 1, True)
                else:
                    (xc,) = ops.TimeSplit.apply(x, 1, True)
                    x = xc
            if not feed_state_forward:
                # RAM-Net: the shared state is updated; the ENCODER feature x feeds the next scale (statenet.py:215-237)
                if branch and i < n - 1:
                    main, side = torch.cuda.current_stream(), ops.branch_stream(x.device, i)
                    side.wait_stream(main)                                  # x_i (and the state) are ready
                    with torch.cuda.stream(side):
                        _, super_state = combs[i](xc, prev_super_state[i], None if out is None else out[i])
                    if not torch.cuda.is_current_stream_capturing():       # (a capture orders its private pool by the graph's own edges;
                        x.record_stream(side)                               # record_stream there left later captures crashing at replay)
                        for t in (prev_super_state[i] if isinstance(prev_super_state[i], (list, tuple)) else (prev_super_state[i],)):
                            t.record_stream(side)
                        for t in (super_state if isinstance(super_state, (list, tuple)) else (super_state,)):
                            t.record_stream(main)                           # consumed on the caller's stream after the join
                    joins.append(side)
                else:
                    _, super_state = combs[i](xc, prev_super_state[i], None if out is None else out[i])      # convlstm: h and c both from the shared state
                state_comb = super_state
                super_states.append(super_state)
            else:
                # baselines: the recurrent OUTPUT feeds the next encoder (statenet.py:276-283)
                if self.state_combination == 'convlstm':
                    x, state_comb = combs[i](x, prev_states_lstm['state_comb'][i])
                else:
                    x, state_comb = combs[i](x, prev_super_state[i])
                super_states.append(x)
            states_lstm['encoders'].append(enc_state)
            states_lstm['state_comb'].append(state_comb)
        for side in joins:
            torch.cuda.current_stream().wait_stream(side)
        return super_states, states_lstm

    def forward_events(self, x, prev_super_state, prev_states_lstm, times=None, out=None):
        return self._encode(x, self.head_events, self.encoders_events, self.state_combination_events,
                            prev_super_state, prev_states_lstm, False, out)

    def forward_images(self, x, prev_super_state, prev_states_lstm, times=None, out=None):
        return self._encode(x, self.head_rgb, self.encoders_rgb, self.state_combination_images,
                            prev_super_state, prev_states_lstm, bool(self.baseline), out)

    # ---------------------------------------------------------------------------------------------- decoder
    def forward_decoder(self, super_states, si=None):
        """si (training step of this package's own trainer): (weight, n_lambda, [target per equal batch segment]) — the scale-invariant loss
        of the decoded maps is formed inside the prediction layer's launches (ops.PredSigmoidSI) and returned beside the prediction:
        (pred, [loss per segment]).  Default: the prediction alone, as in the reference (statenet.py:290-315)."""
        pair = (not bool(self.baseline)) and self.state_combination == 'convlstm'
        pick = (lambda s: s[0]) if pair else (lambda s: s)
        x = pick(super_states[-1])
        for rb in self.resblocks:
            x = rb(x)
        for i, dec in enumerate(self.decoders):
            x = dec(x) if i == 0 else dec(x, pick(super_states[self.num_encoders - i - 1]))   # no skip into decoder 0
        if self.norm in ('BN', 'IN'):          # conv1x1 -> norm -> sigmoid (statenet.py:116-117, 313)
            assert si is None, "fused SI loss: plain prediction layer only"
            return self.pred(x, act='sigmoid').permute(0, 3, 1, 2)           # NHWC [B,H,W,1] -> NCHW view
        if si is not None:
            # (x = the last decoder's ReLU output, consumed here and nowhere else: the prediction layer's backward applies that ReLU's mask to dx)
            pm = torch.is_grad_enabled() and ops.premask_relu_feature(x)
            out = ops.PredSigmoidSI.apply(x, self.pred.conv2d.weight, self.pred.conv2d.bias, float(si[0]), float(si[1]), pm, *si[2])
            return out[0], list(out[1:])
        return ops.PredSigmoid.apply(x, self.pred.conv2d.weight, self.pred.conv2d.bias)
