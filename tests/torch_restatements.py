"""Plain-torch restatements of the HIP pack / unpack kernels of the folded upsample-conv decoders (test references only: the product
path runs ramnet_pack_weight_fold_wino[_dgrad], ramnet_pack_border_weights and ramnet_fold_unpack_wgrad; nothing under rpg_ramnet_amd/
imports this module)."""
import torch

from rpg_ramnet_amd import ops
from rpg_ramnet_amd.ops import FOLD_A, _const, _fold_pair, fold_weights

FOLD_LOST = [[(0, 1), (0,)], [(4,), (3, 4)]]    # [side: near / far edge][slot: distance into the band] -> taps outside the image

# Winograd F(2x2,4x4) of the four parity filters (csrc/conv_wino24.hip; Toom-Cook points 0, 1, -1, 2, inf)
W24_G = [[0.5, 0, 0, 0], [-0.5, -0.5, -0.5, -0.5], [-1 / 6, 1 / 6, -1 / 6, 1 / 6], [1 / 6, 1 / 3, 2 / 3, 4 / 3], [0, 0, 0, 1]]
W24_BT = [[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]]
W24_AT = [[1, 1, 1, 1, 0], [0, 1, -1, 2, 1]]


def fold_weights_wino(w):
    """OIHW 5x5 -> U[class = py*2+px][pos = a*5+b][Cin][Cout] = G W4 G^T of the four 4x4 parity filters (float64)."""
    G = _const("W24_G", W24_G, w.device, torch.float64)
    u = torch.einsum("at,bs,oipqts->pqabio", G, G, fold_weights(w))
    return u.reshape(4, 25, w.shape[1], w.shape[0])


def pack_fold_wino(w):
    """fold_weights_wino() in the lane order of conv_wino24_kernel's B operand (layout: include/ramnet_hip.h)."""
    Cout, Cin = w.shape[0], w.shape[1]
    if _fold_pair(Cout, Cin):       # class = row parity py, the workgroup's 64 columns = (column parity px, 32 channels)
        u = fold_weights_wino(w).float().view(2, 2, 25, Cin, 32).permute(0, 2, 3, 1, 4).reshape(2, 25, Cin // 16, 4, 4, 1, 4, 16)
        return u.permute(0, 2, 5, 1, 6, 3, 7, 4).contiguous().view(-1)
    kc, ncq = (16, 4) if (Cout % 64 == 0 and Cin % 16 == 0) else (8, 2)     # chunk size, 16-channel groups per workgroup
    u = fold_weights_wino(w).float().view(4, 25, Cin // kc, 4, kc // 4, Cout // (16 * ncq), ncq, 16)      # cls pos chunk ks j nb cq l15
    return u.permute(0, 2, 5, 1, 6, 3, 7, 4).contiguous().view(-1)                                       # cls chunk nb pos cq ks l15 j


def pack_fold_wino_dgrad(w):
    """Backward-data of the folded layer on conv_wino24_kernel (RAMNET_IN_PARITY4): Winograd weights of the FLIPPED parity filters
    with the roles of the channels swapped — reduce over (parity class, output channel), produce input channels."""
    Cout, Cin = w.shape[0], w.shape[1]
    G = _const("W24_G", W24_G, w.device, torch.float64)
    u = torch.einsum("at,bs,ncpqts->abpqnc", G, G, fold_weights(w).flip(4, 5)).reshape(1, 25, 4 * Cout, Cin).float()
    u = u.view(1, 25, 4 * Cout // 16, 4, 4, Cin // 64, 4, 16)                   # cls pos chunk ks j nb cq l15
    return u.permute(0, 2, 5, 1, 6, 3, 7, 4).contiguous().view(-1)


def fold_unpack_torch(w4, dU, wr, wc, Cout, Cin, CinWs):
    """Plain-torch statement of ramnet_fold_unpack_wgrad (the product path runs the kernel; tests compare the two): the OIHW 5x5 gradient
    that the workspaces of a folded decoder's backward pass stand for."""
    A = _const("FOLD_A", FOLD_A, w4.device, torch.float32)                        # [p][t][k]
    d4 = w4.view(2, 2, 4, 4, CinWs, Cout)[:, :, :, :, :Cin]
    if dU is not None:        # dW4 += G^T dU G
        G = _const("W24_G", W24_G, w4.device, torch.float32)
        d4 = d4 + torch.einsum("at,bs,pqabio->pqtsio", G, G, dU.view(2, 2, 5, 5, CinWs, Cout))[:, :, :, :, :Cin]
    g = torch.einsum("ptk,qsl,pqtsio->oikl", A, A, d4).contiguous()
    r = wr.view(2, 5, Cin, 2, Cout)          # [side][kx][ci][slot][co]
    c = wc.view(2, 5, Cin, 2, Cout)          # [side][ky][ci][slot][co]
    for side in range(2):
        for slot in range(2):
            for a in FOLD_LOST[side][slot]:
                g[:, :, a, :].sub_(r[side, :, :, slot, :].permute(2, 1, 0))        # [co][ci][kx]
                g[:, :, :, a].sub_(c[side, :, :, slot, :].permute(2, 1, 0))        # [co][ci][ky]
    return g


def border_matrices(w):
    """(Wrows, Wcols), each [2 sides][5*Cin][2*Cout]: MINUS the sums of the taps the zero padding removes at the image border.
    Rows: K index = (kx, ci), N index = (slot, co), lost direction ky; columns the same with ky <-> kx."""
    Cout, Cin = w.shape[0], w.shape[1]

    def mats(wk):                                                          # wk[co][ci][a][b]: `a` is the lost direction
        out = []
        for side in range(2):
            slots = [-sum(wk[:, :, a, :] for a in FOLD_LOST[side][slot]) for slot in range(2)]          # [co][ci][b]
            out.append(torch.stack(slots, 0).permute(3, 2, 0, 1).reshape(5 * Cin, 2 * Cout))              # [b][ci][slot][co]
        return torch.stack(out, 0).contiguous()
    return mats(w), mats(w.transpose(2, 3))


