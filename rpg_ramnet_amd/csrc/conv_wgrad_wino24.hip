// Backward-weights of the folded upsample-conv (decoders, statenet.py:305-308) in the Winograd F(2x2,4x4) domain on gfx950:
//     dU_cls[pos][ci][co] = sum_tiles V_cls[pos][tile][ci] * Z_cls[pos][tile][co],   V = B^T d B,   Z = A g A^T
// with d the 5x5 window of the replicate-padded low-res input of parity class cls = (py, px) and g the 2x2 tile of that
// parity's sub-grid of dy (* ReLU mask); the 4x4 parity-filter gradient is dW4 = G^T dU G (ops.ConvParam._finalize_fold).
// 25 GEMMs per class with M = input channel, N = output channel, K = tiles on v_mfma_f32_32x32x2_f32: 25 instead of 64
// multiplies per tile and channel pair (csrc/conv_wino24.hip has the forward and the matrices).
//
// A workgroup (8 waves) owns 32 x 64 channels of one class and walks batches of 8 tiles: every thread loads the window of one
// (tile, input channel) and the 2x2 gradients of one (tile, output channel) straight from global memory (lanes along channels:
// coalesced), transforms them in registers and writes V[25][8][32] / Z[25][8][64] to LDS (double-buffered, one barrier per
// batch).  Wave (position group of 7/6/6/6, channel half) accumulates 32 x 32 per position (112 VGPRs) over its whole tile range;
// loads and transforms of the next batches are issued between the MFMAs; tile splits meet by atomics in the [4][25][Cin][Cout]
// workspace; the bias gradient rides along.
#include <stdlib.h>
#include "common.hpp"

namespace ramnet {

constexpr int G24_T = 8;                          // tiles per batch
// channels per workgroup: 32 input x 64 output, or (WCI, for 32-output-channel layers) 64 input x 32 output

struct Wgrad24Params {
    const float *x, *g, *gm;
    float *dw, *dbias;
    int ldx, ldg, ldgm, Hp, Wp, H2, W2, Hc, Wc, Cin, Cout, tiles_x, tiles_y, ntiles, nbatch, ncob;
    size_t xbytes, gbytes, gmbytes;       // extents of x / g / gm (buffer addressing: < WOOB, checked on the host)
};

template <bool GM, bool WCI>
__global__ void __launch_bounds__(512, 1) conv_wgrad_wino24_kernel(const Wgrad24Params p) {
    constexpr int G24_CI = WCI ? 64 : 32, G24_CO = WCI ? 32 : 64;
    constexpr int G24_V = 25 * G24_T * G24_CI, G24_Z = 25 * G24_T * G24_CO;      // 6400 / 12800 floats (or swapped)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *V = smem;                   // [2][25][8][CI]
    float *Z = smem + 2 * G24_V;       // [2][25][8][CO]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int pg = wave >> 1, ch = wave & 1;                  // position group, half of the wider channel dimension
    const int p0 = pg == 0 ? 0 : 1 + 6 * pg;                  // positions p0 .. p0 + (pg == 0 ? 7 : 6) - 1
    const int cls = blockIdx.z / p.ncob, cob = blockIdx.z % p.ncob;
    const int py = cls >> 1, px = cls & 1;
    const int c0 = blockIdx.y * G24_CI, n0 = cob * G24_CO;
    const bool vrole = WCI || wave < 4, zrole = !WCI || wave < 4;      // the narrower operand has 256 items per batch, the wider 512
    const int vt = (tid / G24_CI) & 7, vc = tid % G24_CI;      // input item: (tile of the batch, input channel)
    const int zt = (tid / G24_CO) & 7, zc = tid % G24_CO;      // gradient item: (tile, output channel)

    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float vraw[25], zraw[4];
    float bsum = 0.f;
    // Loaders (round 4: 670 -> see the header; two run-time integer divisions, 64-bit pointer arithmetic, ten clamps and a select per value
    // before): the (image, tile row, tile column) of the thread's tile is divided out ONCE and then walks by the pre-divided grid stride
    // with carries; a value is one buffer load at lane offset + scalar offset.  The window of a tile that overhangs the grid, or lies past
    // the last tile, reads whatever finite data (or the zeros past the tensor) the offset finds: it only meets gradient entries that are
    // zero — the Winograd row / column that contains window line 4 multiplies A g's last line, which is the out-of-grid gradient itself.
    struct Walk { int tx, ty, b, T; };
    const int dT = G24_T * (int)gridDim.x;
    const int d_tx = dT % p.tiles_x, d_ty = (dT / p.tiles_x) % p.tiles_y, d_b = (dT / p.tiles_x) / p.tiles_y;
    auto walk_init = [&](Walk &w, int batch, int tile) {
        w.T = batch * G24_T + tile;
        int t = w.T;
        w.tx = t % p.tiles_x;
        t /= p.tiles_x;
        w.ty = t % p.tiles_y, w.b = t / p.tiles_y;
    };
    auto walk_step = [&](Walk &w, int adv) {            // adv = 1: the next batch of this workgroup, 0: the same one again (clamped tail)
        w.T += adv * dT;
        w.tx += adv * d_tx;
        const int cx = w.tx >= p.tiles_x ? 1 : 0;
        w.tx -= cx * p.tiles_x;
        w.ty += adv * d_ty + cx;
        const int cy = w.ty >= p.tiles_y ? 1 : 0;
        w.ty -= cy * p.tiles_y;
        w.b += adv * d_b + cy;
    };
    Walk wv, wz;
    int lbv = 0, lbz = 0;                               // batch the walks stand at
    const auto rx = wino_rsrc(p.x, (unsigned)min((size_t)p.xbytes, (size_t)WOOB));
    const auto rg = wino_rsrc(p.g, (unsigned)min((size_t)p.gbytes, (size_t)WOOB));
    const auto rgm = GM ? wino_rsrc(p.gm, (unsigned)min((size_t)p.gmbytes, (size_t)WOOB)) : rg;
    const bool vch_ok = c0 + vc < p.Cin, zch_ok = n0 + zc < p.Cout;
    const unsigned zch = (unsigned)(n0 + zc);
    const int rowB = p.Wp * p.ldx * 4, colB = p.ldx * 4;                              // bytes per window row / column
    auto load_v = [&](int batch) {
        walk_step(wv, batch != lbv ? 1 : 0);
        lbv = batch;
        const unsigned base = (unsigned)(((wv.b * p.Hp + 2 * wv.ty + py) * p.Wp + 2 * wv.tx + px) * p.ldx + c0 + vc) * 4u;
        const unsigned lane_off = ((wv.T < p.ntiles) & vch_ok) ? base : WOOB;
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c)
                vraw[r * 5 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(lane_off + (unsigned)(r * rowB)), c * colB, 0));
    };
    auto load_z = [&](int batch, bool count) {
        walk_step(wz, batch != lbz ? 1 : 0);
        lbz = batch;
        const bool tok = (wz.T < p.ntiles) & zch_ok;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int oy = 2 * wz.ty + a, ox = 2 * wz.tx + c;
                const bool ok = tok & (oy < p.Hc) & (ox < p.Wc);
                const unsigned pix = (unsigned)((wz.b * p.H2 + 2 * oy + py) * p.W2 + 2 * ox + px);
                float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)(ok ? (pix * p.ldg + zch) * 4u : WOOB), 0, 0));
                if (GM) {
                    const float m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rgm, (int)(ok ? (pix * p.ldgm + zch) * 4u : WOOB), 0, 0));
                    v = m > 0.f ? v : 0.f;
                }
                zraw[a * 2 + c] = v;
                if (count) bsum += v;
            }
    };
    // B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
    auto v_cols = [&]() {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float d0 = vraw[c], d1 = vraw[5 + c], d2 = vraw[10 + c], d3 = vraw[15 + c], d4 = vraw[20 + c];
            vraw[c] = 2.f * (d0 - d2) - d1 + d3;
            vraw[5 + c] = d3 - d2 - 2.f * d1;
            vraw[10 + c] = 2.f * d1 - 3.f * d2 + d3;
            vraw[15 + c] = d3 - d1;
            vraw[20 + c] = 2.f * (d1 - d3) - d2 + d4;
        }
    };
    auto v_row = [&](float *vbuf, int i) {
        const float d0 = vraw[i * 5], d1 = vraw[i * 5 + 1], d2 = vraw[i * 5 + 2], d3 = vraw[i * 5 + 3], d4 = vraw[i * 5 + 4];
        float *dst = vbuf + ((i * 5) * G24_T + vt) * G24_CI + vc;
        dst[0 * G24_T * G24_CI] = 2.f * (d0 - d2) - d1 + d3;
        dst[1 * G24_T * G24_CI] = d3 - d2 - 2.f * d1;
        dst[2 * G24_T * G24_CI] = 2.f * d1 - 3.f * d2 + d3;
        dst[3 * G24_T * G24_CI] = d3 - d1;
        dst[4 * G24_T * G24_CI] = 2.f * (d1 - d3) - d2 + d4;
    };
    // Z = A g A^T, A = [1 0; 1 1; 1 -1; 1 2; 0 1]: row i of Z from w_i = (g0*, g0*+g1*, g0*-g1*, g0*+2 g1*, g1*)
    auto z_row = [&](float *zbuf, int i) {
        float w0, w1;                                   // (A g)[i][0], (A g)[i][1]
        if (i == 0) w0 = zraw[0], w1 = zraw[1];
        if (i == 1) w0 = zraw[0] + zraw[2], w1 = zraw[1] + zraw[3];
        if (i == 2) w0 = zraw[0] - zraw[2], w1 = zraw[1] - zraw[3];
        if (i == 3) w0 = zraw[0] + 2.f * zraw[2], w1 = zraw[1] + 2.f * zraw[3];
        if (i == 4) w0 = zraw[2], w1 = zraw[3];
        float *dst = zbuf + ((i * 5) * G24_T + zt) * G24_CO + zc;
        dst[0 * G24_T * G24_CO] = w0;
        dst[1 * G24_T * G24_CO] = w0 + w1;
        dst[2 * G24_T * G24_CO] = w0 - w1;
        dst[3 * G24_T * G24_CO] = w0 + 2.f * w1;
        dst[4 * G24_T * G24_CO] = w1;
    };

    const int step = gridDim.x;
    int batch = blockIdx.x;
    if (batch < p.nbatch) {
        const int last = batch + ((p.nbatch - 1 - batch) / step) * step;
        // prologue: batch -> buffer 0; raw data of the next batch in registers
        walk_init(wv, batch, vt), walk_init(wz, batch, zt);
        lbv = lbz = batch;
        if (zrole) load_z(batch, true);
        if (vrole) load_v(batch);
        if (zrole) {
#pragma unroll
            for (int i = 0; i < 5; ++i) z_row(Z, i);
        }
        if (vrole) {
            v_cols();
#pragma unroll
            for (int i = 0; i < 5; ++i) v_row(V, i);
        }
        {
            const bool more = batch + step <= last;
            if (zrole) load_z(min(batch + step, last), more);
            if (vrole) load_v(min(batch + step, last));
        }
        __syncthreads();
        const int aoff = (kk * 4) * G24_CI + (WCI ? ch * 32 : 0) + l31, boff = (kk * 4) * G24_CO + (WCI ? 0 : ch * 32) + l31;
        int buf = 0;
        for (; batch <= last; batch += step, buf ^= 1) {
            const float *vb = V + buf * G24_V, *zb = Z + buf * G24_Z;
            float *vn = V + (buf ^ 1) * G24_V, *zn = Z + (buf ^ 1) * G24_Z;
            const int b2 = min(batch + 2 * step, last);
            const bool more2 = batch + 2 * step <= last;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if (q < 6 || pg == 0) {
                    const int pos = p0 + q;
                    const float *ap = vb + pos * (G24_T * G24_CI) + aoff, *bp = zb + pos * (G24_T * G24_CO) + boff;
                    const float a0 = ap[0], a1 = ap[G24_CI], a2 = ap[2 * G24_CI], a3 = ap[3 * G24_CI];
                    const float b0 = bp[0], b1 = bp[G24_CO], b2v = bp[2 * G24_CO], b3 = bp[3 * G24_CO];
                    __builtin_amdgcn_sched_barrier(0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // raw data of the next batch (in registers) -> the other LDS buffer; then the loads of the batch after it
                    if (q == 0 && zrole) z_row(zn, 0), z_row(zn, 1), z_row(zn, 2);
                    if (q == 1 && zrole) z_row(zn, 3), z_row(zn, 4);
                    if (q == 2 && vrole) v_cols();
                    if (q == 3 && vrole) v_row(vn, 0), v_row(vn, 1), v_row(vn, 2);
                    if (q == 4 && vrole) v_row(vn, 3), v_row(vn, 4);
                    if (q == 4) {
                        if (zrole) load_z(b2, more2);
                        if (vrole) load_v(b2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2v, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
    }

    // D[row = input channel][col = output channel] of position p0 + q -> dw[((cls*25 + pos)*Cin + c)*Cout + n]
    const int n = n0 + (WCI ? 0 : ch * 32) + l31;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        if (q < 6 || pg == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (WCI ? ch * 32 : 0) + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (c < p.Cin && n < p.Cout) atomicAdd(p.dw + (((size_t)cls * 25 + p0 + q) * p.Cin + c) * p.Cout + n, acc[q][r]);
            }
        }
    }
    if (p.dbias != nullptr && blockIdx.y == 0) {
        __syncthreads();
        float *red = smem;                        // [8][CO]
        if (zrole) red[zt * G24_CO + zc] = bsum;
        __syncthreads();
        if (tid < G24_CO) {
            float t = 0.f;
            for (int g = 0; g < 8; ++g) t += red[g * G24_CO + tid];
            if (n0 + tid < p.Cout) atomicAdd(p.dbias + n0 + tid, t);
        }
    }
}

int launch_wgrad_wino24(const ramnet_wgrad_desc &d, hipStream_t st) {
    // x0 = replicate-padded low-res input [B][Hin = H+4][Win = W+4][C0]; dout / gmask = [B][HoG = 2H][WoG = 2W][Cout]; Ho, Wo = H, W
    const bool wci = d.Cout % 64 != 0;           // 32-output-channel layers: 64 x 32 channels per workgroup
    const int G24_CI = wci ? 64 : 32, G24_CO = wci ? 32 : 64;
    RAMNET_CHECK_ARG(d.in_mode == RAMNET_IN_PLAIN && d.stride == 1 && d.C0 % G24_CI == 0 && d.Cout % 32 == 0);
    RAMNET_CHECK_ARG(d.Hin == d.Ho + 4 && d.Win == d.Wo + 4 && d.HoG == 2 * d.Ho && d.WoG == 2 * d.Wo && d.Ho >= 2 && d.Wo >= 2);
    Wgrad24Params q;
    q.x = d.x0, q.g = d.dout, q.gm = d.gmask, q.dw = d.dw, q.dbias = d.dbias;
    q.ldx = d.ld0, q.ldg = d.ldg, q.ldgm = d.ldgm, q.Hp = d.Hin, q.Wp = d.Win, q.H2 = d.HoG, q.W2 = d.WoG, q.Hc = d.Ho, q.Wc = d.Wo;
    q.Cin = d.C0, q.Cout = d.Cout;
    q.xbytes = (size_t)d.B * d.Hin * d.Win * d.ld0 * 4, q.gbytes = (size_t)d.B * d.HoG * d.WoG * d.ldg * 4;
    q.gmbytes = d.gmask ? (size_t)d.B * d.HoG * d.WoG * d.ldgm * 4 : 0;
    // per-lane 32-bit byte offsets; a tile past the last one may stand up to one grid stride of images beyond the tensor
    RAMNET_CHECK_ARG(2 * q.xbytes < WOOB && 2 * q.gbytes < WOOB && 2 * q.gmbytes < WOOB);
    q.tiles_x = cdiv(d.Wo, 2), q.tiles_y = cdiv(d.Ho, 2), q.ntiles = q.tiles_x * q.tiles_y * d.B, q.nbatch = cdiv(q.ntiles, G24_T);
    q.ncob = cdiv(d.Cout, G24_CO);
    const int gy = d.C0 / G24_CI, gz = q.ncob * 4;
    int splits = 256 / (gy * gz);                   // (128 / 384 workgroups measured the same training step)
    if (splits > q.nbatch) splits = q.nbatch;
    if (splits < 1) splits = 1;
    const size_t lds = (size_t)2 * 25 * G24_T * (32 + 64) * sizeof(float);
    const dim3 grid(splits, gy, gz);
    auto go = [&](auto kern) -> int {
        RAMNET_FULL_LDS((kern));
        note_kernel("conv_wgrad_wino24_kernel<%d,%d>", d.gmask ? 1 : 0, (int)wci);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, q);
        return 0;
    };
    const int rc = wci ? (d.gmask ? go(conv_wgrad_wino24_kernel<true, true>) : go(conv_wgrad_wino24_kernel<false, true>))
                       : (d.gmask ? go(conv_wgrad_wino24_kernel<true, false>) : go(conv_wgrad_wino24_kernel<false, false>));
    if (rc) return rc;
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet
