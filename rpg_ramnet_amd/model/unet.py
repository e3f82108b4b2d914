"""Non-recurrent UNet (ERGB2Depth baseline) on HIP kernels; names/order as RAM_Net/model/unet.py:87-130."""
import torch.nn as nn

from .. import ops
from .submodules import ConvLayer, UpsampleConvLayer, TransposedConvLayer, ResidualBlock


class UNet(nn.Module):
    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', activation='sigmoid',
                 num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None, use_upsample_conv=True):
        super().__init__()
        if skip_type not in ('sum', 'concat', 'no_skip', None):
            raise KeyError('Could not identify skip_type, please add "skip_type":'
                           ' "sum", "concat" or "no_skip" to config["model"]')
        if skip_type not in ('sum', 'concat'):
            raise NotImplementedError("UNet skip_type %r does not run in the reference either: its decoders are built 2c wide "
                                      "(unet.py:78) and fed c channels" % (skip_type,))
        assert activation == 'sigmoid' and num_output_channels == 1
        self.num_encoders, self.norm = num_encoders, norm
        self.skip_type = skip_type
        wide = 1 if skip_type == 'sum' else 2          # decoder / pred input width (unet.py:78, :83)
        if use_upsample_conv:
            print('Using UpsampleConvLayer (slow, but no checkerboard artefacts)')
            Up = UpsampleConvLayer
        else:
            print('Using TransposedConvLayer (fast, with checkerboard artefacts)')
            Up = TransposedConvLayer
        max_c = base_num_channels * pow(2, num_encoders)
        enc_in = [base_num_channels * pow(2, i) for i in range(num_encoders)]
        enc_out = [base_num_channels * pow(2, i + 1) for i in range(num_encoders)]
        self.head = ConvLayer(num_input_channels, base_num_channels, kernel_size=5, stride=1, padding=2)
        self.encoders = nn.ModuleList([ConvLayer(i, o, kernel_size=5, stride=2, padding=2, norm=norm)
                                       for i, o in zip(enc_in, enc_out)])
        self.resblocks = nn.ModuleList([ResidualBlock(max_c, max_c, norm=norm) for _ in range(num_residual_blocks)])
        self.decoders = nn.ModuleList([Up(wide * c, c // 2, kernel_size=5, padding=2, norm=norm) for c in reversed(enc_out)])
        self.pred = ConvLayer(wide * base_num_channels, num_output_channels, 1, activation=None, norm=norm)

    def forward(self, x):
        x = self.head(x)
        head, blocks = x, []
        for enc in self.encoders:
            x = enc(x)
            blocks.append(x)
        for rb in self.resblocks:
            x = rb(x)
        for i, dec in enumerate(self.decoders):                      # every decoder gets its skip (unet.py:126-127)
            skip = blocks[self.num_encoders - i - 1]
            if self.skip_type == 'sum':
                x = dec(x, skip)                                     # the sum is fused into the layer's loader
            else:
                x = dec(ops.Concat.apply(x, skip))
        x = ops.Add.apply(x, head) if self.skip_type == 'sum' else ops.Concat.apply(x, head)      # head skip (unet.py:129)
        if self.norm in ('BN', 'IN'):
            return self.pred(x, act='sigmoid').permute(0, 3, 1, 2)           # NHWC [B,H,W,1] -> NCHW view
        return ops.PredSigmoid.apply(x, self.pred.conv2d.weight, self.pred.conv2d.bias)
