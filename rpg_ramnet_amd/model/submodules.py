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


def _check_norm(norm):
    if norm in ("BN", "IN"):
        raise NotImplementedError("norm=%r: no shipped RAM-Net config uses BN/IN (all use 'none'); not on the HIP path" % norm)


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
        _check_norm(norm)
        assert kernel_size in (1, 3, 5) and padding == kernel_size // 2, "HIP path: 'same' 1x1/3x3/5x5 convs"
        assert activation in ('relu', None)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.stride, self.relu = stride, activation == 'relu'

    def cp(self):
        return self._cp("c", [self.conv2d.weight], [self.conv2d.bias])

    def forward(self, x):
        return ops.ConvAct.apply(x, None, self.conv2d.weight, self.conv2d.bias, self.cp(), self.stride, self.relu, False)


class UpsampleConvLayer(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        _check_norm(norm)
        assert kernel_size == 5 and padding == 2 and stride == 1 and activation == 'relu'
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)

    def cp(self):
        return self._cp("c", [self.conv2d.weight], [self.conv2d.bias])

    def forward(self, x, skip=None):
        """relu(conv5x5(bilinear_x2(x [+ skip]))) — upsample and skip sum are fused into the conv's tile loader."""
        return ops.ConvAct.apply(x, skip, self.conv2d.weight, self.conv2d.bias, self.cp(), 1, True, True)


class TransposedConvLayer(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        _check_norm(norm)
        assert kernel_size == 5 and padding == 2 and activation == 'relu'
        self.transposed_conv2d = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=padding,
                                                    output_padding=1, bias=True)

    def forward(self, x, skip=None):
        if skip is not None:
            x = ops.Add.apply(x, skip)
        t = self.transposed_conv2d
        return ops.TConvAct.apply(x, t.weight, t.bias, self._cp("t", [t.weight], [t.bias]))


class ResidualBlock(nn.Module, _Lazy):
    def __init__(self, in_channels, out_channels, norm=None):
        super().__init__()
        _check_norm(norm)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):
        c1 = self._cp("c1", [self.conv1.weight], [self.conv1.bias])
        c2 = self._cp("c2", [self.conv2.weight], [self.conv2.bias])
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
