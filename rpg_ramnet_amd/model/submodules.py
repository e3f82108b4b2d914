"""Layer primitives with the reference's parameter names/shapes/initialisation (so `state_dict`s interchange
and `torch.manual_seed(0)` yields the same weights), computed by HIP kernels.

Reference: RAM_Net/model/submodules.py — ConvLayer :8-35, UpsampleConvLayer :69-97, RecurrentConvLayer :100-120,
Recurrent2ConvLayer :122-142, ResidualBlock :182-215, ConvLSTM :303-358, ConvGRU :414-454.
The nn.Conv2d members are parameter holders only (their forward is never called).  All activations are NHWC.
"""
import torch
import torch.nn as nn
from torch.nn import init

from .. import ops


def _norm_layer(norm, channels, track=True):
    """The reference's norm member (submodules.py:19-24, 55-59, 85-89; ResidualBlock :188-193 builds its InstanceNorm WITHOUT running
    statistics): a parameter / buffer holder with the same state_dict keys; ops.norm_act computes it.  Any other string ('none') = no norm."""
    if norm == 'BN':
        return nn.BatchNorm2d(channels, momentum=0.1)
    if norm == 'IN':
        return nn.InstanceNorm2d(channels, track_running_stats=track)
    return None


class _Lazy:
    """Builds ops.ConvParam on first use (after .to(device))."""

    def _cp(self, name, weights, biases, gates=1):
        d = self.__dict__.setdefault("_cps", {})
        if name not in d:
            d[name] = ops.ConvParam(weights, biases, gates)
        return d[name]


class ConvLayer(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        assert kernel_size in (1, 3, 5) and padding == kernel_size // 2, "HIP path: 'same' 1x1/3x3/5x5 convs"
        assert activation in ('relu', None)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=norm != 'BN')
        self.stride, self.relu, self.norm = stride, activation == 'relu', norm
        if norm in ('BN', 'IN'):
            self.norm_layer = _norm_layer(norm, out_channels)

    def cp(self):
        return self._cp("c", [self.conv2d.weight], [self.conv2d.bias])

    def forward(self, x, act=None):
        """act: activation applied instead of the layer's own (the prediction layer's sigmoid, statenet.py:313)."""
        if self.norm in ('BN', 'IN'):
            c = self.conv2d
            if c.out_channels == 1 and c.kernel_size == (1, 1):       # the prediction layer: its own one-channel kernels
                t = ops.PredLinear.apply(x, c.weight, c.bias)
            else:
                t = ops.ConvAct.apply(x, None, c.weight, c.bias, self.cp(), self.stride, False, False)
            return ops.norm_act(t, self.norm_layer, act or ('relu' if self.relu else None))
        assert act is None
        return ops.ConvAct.apply(x, None, self.conv2d.weight, self.conv2d.bias, self.cp(), self.stride, self.relu, False)


class UpsampleConvLayer(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        assert kernel_size == 5 and padding == 2 and stride == 1 and activation == 'relu'
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=norm != 'BN')
        self.norm = norm
        if norm in ('BN', 'IN'):
            self.norm_layer = _norm_layer(norm, out_channels)

    def cp(self):
        return self._cp("c", [self.conv2d.weight], [self.conv2d.bias])

    def forward(self, x, skip=None):
        """relu([norm](conv5x5(bilinear_x2(x [+ skip])))) — upsample and skip sum are fused into the conv's tile loader."""
        if self.norm in ('BN', 'IN'):
            t = ops.ConvAct.apply(x, skip, self.conv2d.weight, self.conv2d.bias, self.cp(), 1, False, True)
            return ops.norm_act(t, self.norm_layer, 'relu')
        return ops.ConvAct.apply(x, skip, self.conv2d.weight, self.conv2d.bias, self.cp(), 1, True, True)


class TransposedConvLayer(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        assert kernel_size == 5 and padding == 2 and activation == 'relu'
        self.transposed_conv2d = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=padding,
                                                    output_padding=1, bias=norm != 'BN')
        self.norm = norm
        if norm in ('BN', 'IN'):
            self.norm_layer = _norm_layer(norm, out_channels)

    def forward(self, x, skip=None):
        if skip is not None:
            x = ops.Add.apply(x, skip)
        t = self.transposed_conv2d
        cp = self._cp("t", [t.weight], [t.bias])
        if self.norm in ('BN', 'IN'):
            return ops.norm_act(ops.TConvAct.apply(x, t.weight, t.bias, cp, False), self.norm_layer, 'relu')
        return ops.TConvAct.apply(x, t.weight, t.bias, cp)


class ResidualBlock(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, norm=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=norm != 'BN')
        self.norm = norm
        if norm in ('BN', 'IN'):        # (construction order of submodules.py:186-196: conv1, bn1, bn2, conv2)
            self.bn1 = _norm_layer(norm, out_channels, track=False)
            self.bn2 = _norm_layer(norm, out_channels, track=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=norm != 'BN')

    def forward(self, x):
        c1 = self._cp("c1", [self.conv1.weight], [self.conv1.bias])
        c2 = self._cp("c2", [self.conv2.weight], [self.conv2.bias])
        if self.norm in ('BN', 'IN'):
            t = ops.ConvAct.apply(x, None, self.conv1.weight, self.conv1.bias, c1, 1, False, False)
            t = ops.norm_act(t, self.bn1, 'relu')
            t = ops.ConvAct.apply(t, None, self.conv2.weight, self.conv2.bias, c2, 1, False, False)
            return ops.norm_act(t, self.bn2, 'relu', res=x)
        t = ops.ConvAct.apply(x, None, self.conv1.weight, self.conv1.bias, c1, 1, True, False)
        return ops.ResConv.apply(t, x, self.conv2.weight, self.conv2.bias, c2)


class ConvLSTM(nn.Module, _Lazy):
    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        assert kernel_size == 3 and input_size == hidden_size
        self.input_size, self.hidden_size = input_size, hidden_size
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=1)

    def forward(self, x, prev_state=None, out=None):
        """out: optional (h, c) NHWC buffers that receive the new state (streaming runtimes)."""
        if prev_state is None:
            z = torch.zeros_like(x)
            prev_state = (z, z)
        cp = self._cp("g", [self.Gates.weight], [self.Gates.bias], gates=4)
        oh, oc = out if out is not None else (None, None)
        h, c = ops.LSTMCell.apply(x, prev_state[0], prev_state[1], self.Gates.weight, self.Gates.bias, cp, oh, oc)
        return h, c


class ConvGRU(nn.Module, _Lazy):
    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        assert kernel_size == 3 and input_size == hidden_size
        self.input_size, self.hidden_size = input_size, hidden_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=1)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=1)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=1)
        init.orthogonal_(self.reset_gate.weight)
        init.orthogonal_(self.update_gate.weight)
        init.orthogonal_(self.out_gate.weight)
        init.constant_(self.reset_gate.bias, 0.)
        init.constant_(self.update_gate.bias, 0.)
        init.constant_(self.out_gate.bias, 0.)

    def forward(self, x, prev_state, out=None):
        """out: optional NHWC buffer that receives the new state (streaming runtimes)."""
        if prev_state is None:
            prev_state = torch.zeros_like(x)
        u, r, o = self.update_gate, self.reset_gate, self.out_gate
        cp_ur = self._cp("ur", [u.weight, r.weight], [u.bias, r.bias])
        cp_o = self._cp("o", [o.weight], [o.bias])
        return ops.GRUCell.apply(x, prev_state, u.weight, u.bias, r.weight, r.bias, o.weight, o.bias, cp_ur, cp_o, out)


class RecurrentConvLayer(nn.Module):
    """State-combination block: the recurrent cell alone, kernel forced to 3 (submodules.py:112-114)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, recurrent_block_type='convlstm',
                 activation='relu', norm=None):
        super().__init__()
        assert recurrent_block_type in ['convlstm', 'convgru']
        self.recurrent_block_type = recurrent_block_type
        block = ConvLSTM if recurrent_block_type == 'convlstm' else ConvGRU
        self.recurrent_block = block(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def forward(self, x, prev_state, out=None):
        state = self.recurrent_block(x, prev_state) if out is None else self.recurrent_block(x, prev_state, out)
        x = state[0] if self.recurrent_block_type == 'convlstm' else state
        return x, state


class Recurrent2ConvLayer(nn.Module):
    """Encoder with its own recurrence: strided conv then the recurrent cell (submodules.py:122-142)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, recurrent_block_type='convlstm',
                 activation='relu', norm=None):
        super().__init__()
        assert recurrent_block_type in ['convlstm', 'convgru']
        self.recurrent_block_type = recurrent_block_type
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, padding, activation, norm)
        block = ConvLSTM if recurrent_block_type == 'convlstm' else ConvGRU
        self.recurrent_block = block(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def forward(self, x, prev_state):
        x = self.conv(x)
        state = self.recurrent_block(x, prev_state)
        x = state[0] if self.recurrent_block_type == 'convlstm' else state
        return x, state
