"""rpg_ramnet_amd — MI355X (gfx950) native RAM-Net hot path behind the reference's module API.

    from rpg_ramnet_amd.model.model import ERGB2DepthRecurrent, ERGB2Depth   # drop-in for RAM_Net/model/model.py

Compute runs in hand-written HIP kernels (rpg_ramnet_amd/csrc, C ABI in include/ramnet_hip.h); there is no CPU
or eager-PyTorch fallback — a missing librpg_ramnet_hip.so raises at the first kernel call.
"""
__version__ = "0.1.0"
