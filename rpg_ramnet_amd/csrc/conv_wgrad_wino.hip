// Winograd F(2x2, 3x3) backward-weights for gfx950 (MI355X): the 3x3 stride-1 layers of the RAM-Net path (ConvGRU gates /
// candidate, residual blocks; the stride-2 5x5 encoders through their space-to-depth view), exact-fp32 arithmetic on
// v_mfma_f32_32x32x2_f32.
//
//   y = A^T [ (G g G^T) .* (B^T d B) ] A   =>   dU_p[ci][co] = sum_tiles V_p[tile][ci] * Z_p[tile][co],   V = B^T d B,
//   Z = A dy A^T (the 2x2 output-gradient tile spread to 4x4),   dg = G^T dU G  (ramnet_unpack_wgrad_wino).
//
// 16 independent GEMMs, M = input channel, N = output channel, K = Winograd tiles: 16 MACs per tile and channel pair
// instead of the 36 of the direct form (2.25x fewer MFMAs).  A workgroup (4 waves) owns 32 input x 64 output channels and
// walks batches of 8 tiles (a 2 x 16 or an 8 x 4 pixel strip of the output); the four waves split the ROWS of the 4 x 4 transform
// grid: wave w accumulates positions 4w .. 4w+3 (4 positions x 2 output-channel blocks x 32x32 = 128 accumulator VGPRs) over its
// whole tile range.  One MFMA step reduces over two tiles: lane (channel = lane & 31, tile parity = lane >> 5) reads its tile's
// raw input window rows / gradient pixels from LDS (lanes along channels: conflict-free 4-byte reads), forms row w of B^T d B
// resp. A dy A^T in registers — and those ARE the A / B operands of the MFMA (K index = tile parity).  Nothing transformed goes
// through LDS; the raw strips (same fused loaders as everywhere: concatenation, h*r, ReLU mask, space-to-depth view) are
// double-buffered with ONE barrier per batch (behind tile pair 2, so that the first pair of the next batch is prepared under
// pair 3), and the transform of tile pair s+1 sits in the gaps between the MFMAs of tile pair s.  Partial sums of the tile splits meet in a [16][Cin][Cout] fp32 workspace through coalesced atomic adds; the bias
// gradient (sum of dy) rides along.
// Measured against the previous formulation (V[16][ci][8] / Z[16][64][8] in LDS, a transform phase between two barriers per
// batch; same-box A/B, round 2): the six ConvGRU backward-weights launches of a pass 1.596 -> 1.468 ms.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

namespace ramnet {

// output channels per workgroup = 32 * NF: NF = 2 (two workgroups per CU) or 4 (one per CU, 256 accumulator VGPRs: every
// transformed input row then feeds twice the MFMAs)

constexpr int WGRAD_WINO_TARGET = 384;      // workgroups per launch the tile splits aim at

struct WgradWinoParams {
    InSrc src;
    int bx_n, ty_n, nbatch;     // batches per row, tile rows per image, total
    int dy0, dx0;               // offset of the first filter tap
};

// XMK: second operand of the input loader — 0 none, 1 ReLU mask (x * (xm > 0)), 2 product (the h*r half of a CAT_MUL input: the
// workgroups of the plain half load zeros from an out-of-range offset and scale by 1); GM: ReLU mask on dy.  The raw strips are
// fetched with buffer loads: a pixel outside the image is an out-of-range offset (returns 0), a thread without a slot stores into
// a scratch cell — the staging slices between the MFMAs carry no branch and no select besides that offset.
// TXB = tiles per batch row: 8 (a 2 x 16 pixel strip) or 2 (8 x 4 pixels: the 43- / 86-pixel-wide maps of the coarse scales then
// lose 2 % instead of 10 % of the work to the partial last strip).
template <int TXB, int NF = 2> struct GrGeom {
    static constexpr int GW_CO = 32 * NF;
    static constexpr int TYB = 8 / TXB, YH = 2 * TYB, YW = 2 * TXB, PH = YH + 2, PW = YW + 2;
    static constexpr int XPIX = PH * PW, XSLOTS = XPIX * 8, NXS = (XSLOTS + 255) / 256;
    static constexpr int XP = XPIX * 32;           // raw input strip [PH x PW pixels][32 channels]
    static constexpr int YP = 32 * GW_CO;          // raw gradient strip [YH x YW = 32 pixels][64 / 128 channels]
    static constexpr int SX = TXB == 8 ? 4 * 32 : 2 * PW * 32, SY = TXB == 8 ? 4 : 2 * YW;     // step of a tile pair (floats / pixels)
};

template <int XMK, bool GM, int TXB, int NF>
__global__ void __launch_bounds__(256, NF == 4 ? 1 : NF == 1 ? 3 : 2) conv_wgrad_wino_r_kernel(const ramnet_wgrad_desc p, const WgradWinoParams q) {
    using G = GrGeom<TXB, NF>;
    constexpr int NT = 256, XQ = 8, GW_CI = 32, NXS = G::NXS, NYS = NF, XSLOTS = G::XSLOTS, GR_XP = G::XP, GR_YP = G::YP;
    constexpr int GW_CO = G::GW_CO, YQ = GW_CO / 4;          // channel quads per gradient pixel
    constexpr int PW = G::PW, YW = G::YW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xp = smem;                   // [2][72][32]
    float *Yp = smem + 2 * GR_XP;       // [2][32][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int c0 = blockIdx.y * GW_CI, n0 = blockIdx.z * GW_CO;
    const InSrc &s = q.src;

    f32x16 acc[4][NF];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][f][r] = 0.f;

    // ---- raw-data prefetch: per-thread slot geometry, wave-uniform base pointers, buffer resources rebuilt per batch
    float4 xr[NXS], xm[NXS], yr[NYS], ym[NYS];
    const bool second = (s.mode == RAMNET_IN_CAT || s.mode == RAMNET_IN_CAT_MUL) && c0 >= s.C0;
    const bool use_m = XMK == 1 || (XMK == 2 && second);
    const float m_one = use_m ? 0.f : 1.f;      // XMK == 2, plain half: r * (0 + 1)
    const bool s2d = s.mode == RAMNET_IN_S2D;
    const int sgrp = s2d ? c0 >> s.ld1 : 0;
    const int rowS = s2d ? 4 * s.Win : s.Win, colS = s2d ? 2 : 1;
    const float *xsrc = second ? s.x1 + (c0 - s.C0) : s2d ? s.x0 + ((sgrp >> 1) * 2 * s.Win + (sgrp & 1)) * s.ld0 + (c0 - (sgrp << s.ld1)) : s.x0 + c0;
    const float *msrc = s.mode == RAMNET_IN_RELUMASK ? s.xm + c0 : s.xm + (c0 - s.C0);
    const int ldS = second ? s.ld1 : s.ld0;
    int xpy[NXS], xpx[NXS], ypx[NYS], ypy[NYS], xdst[NXS];
    unsigned xoff[NXS], xmoff[NXS], yoff[NYS], ymoff[NYS];
    bool xslot[NXS], yslot[NYS];
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
        const int sl = tid + i * NT, qd = sl % XQ, pix = sl / XQ;
        xpy[i] = pix / PW, xpx[i] = pix - xpy[i] * PW;
        xslot[i] = sl < XSLOTS && c0 + qd * 4 < s.Cin;
        xoff[i] = (unsigned)((xpy[i] * rowS + xpx[i] * colS) * ldS + qd * 4) * 4u;
        xmoff[i] = (unsigned)((xpy[i] * s.Win + xpx[i]) * s.ldm + qd * 4) * 4u;
        xdst[i] = sl < XSLOTS ? (sl / XQ) * 32 + (sl % XQ) * 4 : -1;
    }
#pragma unroll
    for (int i = 0; i < NYS; ++i) {
        const int sl = tid + i * NT, qd = sl % YQ, pix = sl / YQ;
        ypx[i] = pix % YW, ypy[i] = pix / YW;
        yslot[i] = n0 + qd * 4 < p.Cout;
        yoff[i] = (unsigned)((ypy[i] * p.Wo + ypx[i]) * p.ldg + n0 + qd * 4) * 4u;
        ymoff[i] = (unsigned)((ypy[i] * p.Wo + ypx[i]) * p.ldgm + n0 + qd * 4) * 4u;
    }
    int lb_ty = 0, lb_bx = 0;
    // Buffer resources built ONCE: each starts `pad` pixels in front of its tensor (the strip of a top / left batch begins before the
    // image: only out-of-image threads would reach that, and they read WOOB) and spans WOOB bytes; a batch then only moves four scalar
    // byte offsets (32-bit: tensors are < WOOB bytes, checked on the host) instead of rebuilding four descriptors from 64-bit pointers
    const int pad_src = s2d ? -(q.dy0 * (4 * s.Win) + q.dx0 * 2) : -(q.dy0 * s.Win + q.dx0), pad_m = -(q.dy0 * s.Win + q.dx0);
    const int padS = pad_src > 0 ? pad_src : 0, padM = pad_m > 0 ? pad_m : 0;
    unsigned so_x = 0, so_m = 0, so_g = 0, so_gm = 0;
    // batch -> (image, tile row, strip): divided out once; the walk then advances by the (pre-divided) grid stride with carries —
    // a few scalar operations per batch instead of two ~40-instruction integer divisions in front of the loads
    int lb_b = 0, lb_batch = 0;
    const int st_bx = (int)gridDim.x % q.bx_n, st_ty = ((int)gridDim.x / q.bx_n) % q.ty_n, st_b = ((int)gridDim.x / q.bx_n) / q.ty_n;
    auto load_first = [&](int batch) {          // (once, in front of the loop: the two integer divisions)
        int tt = batch;
        lb_bx = tt % q.bx_n;
        tt /= q.bx_n;
        lb_ty = tt % q.ty_n;
        lb_b = tt / q.ty_n;
        lb_batch = batch;
    };
    auto load_begin = [&](int batch) {
        {                                       // batch == lb_batch (first call, clamped tail) or lb_batch + gridDim.x: branch-free
            const int adv = batch != lb_batch ? 1 : 0;
            lb_bx += adv * st_bx;
            const int cx = lb_bx >= q.bx_n ? 1 : 0;
            lb_bx -= cx * q.bx_n;
            lb_ty += adv * st_ty + cx;
            const int cy = lb_ty >= q.ty_n ? 1 : 0;
            lb_ty -= cy * q.ty_n;
            lb_b += adv * st_b + cy;
        }
        lb_batch = batch;
        const int b = lb_b;
        const int pix = (b * p.Ho + G::YH * lb_ty) * p.Wo + YW * lb_bx;
        const int corner = pix + q.dy0 * s.Win + q.dx0;
        const int corner_src = s2d ? (b * p.Ho + G::YH * lb_ty + q.dy0) * rowS + (YW * lb_bx + q.dx0) * colS : corner;
        so_x = (unsigned)((corner_src + padS) * ldS) * 4u;
        if (XMK) so_m = (unsigned)((corner + padM) * s.ldm) * 4u;
        so_g = (unsigned)(pix * p.ldg) * 4u;
        if (GM) so_gm = (unsigned)(pix * p.ldgm) * 4u;
    };
    const auto rx = wino_rsrc(xsrc - (long)padS * ldS, WOOB);
    const auto rmk = XMK ? wino_rsrc(msrc - (long)padM * s.ldm, WOOB) : rx;
    const auto rg = wino_rsrc(p.dout, WOOB);
    const auto rgm = GM ? wino_rsrc(p.gmask, WOOB) : rg;
    auto bload = [](decltype(rx) r, unsigned vo, unsigned so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)so, 0));
    };
    auto load_x = [&](int i) {
        const int iy = G::YH * lb_ty + q.dy0 + xpy[i], ix = YW * lb_bx + q.dx0 + xpx[i];
        const bool ok = xslot[i] & ((unsigned)iy < (unsigned)s.Hin) & ((unsigned)ix < (unsigned)s.Win);      // (no short-circuit: no exec regions)
        xr[i] = bload(rx, ok ? xoff[i] : WOOB, so_x);
        if (XMK) xm[i] = bload(rmk, (ok & use_m) ? xmoff[i] : WOOB, so_m);
    };
    auto load_y = [&](int i) {
        const bool ok = yslot[i] & (G::YH * lb_ty + ypy[i] < p.Ho) & (YW * lb_bx + ypx[i] < p.Wo);
        yr[i] = bload(rg, ok ? yoff[i] : WOOB, so_g);
        if (GM) ym[i] = bload(rgm, ok ? ymoff[i] : WOOB, so_gm);
    };
    auto load_raw = [&](int batch) {
        load_begin(batch);
#pragma unroll
        for (int i = 0; i < NXS; ++i) load_x(i);
#pragma unroll
        for (int i = 0; i < NYS; ++i) load_y(i);
    };
    float4 bsum = f4zero();                      // bias gradient partial of channel quad (tid % YQ)
    float *scratch = smem + 2 * (GR_XP + GR_YP) + tid * 4;          // 256 spare 16-byte cells behind the strips
    auto store_x = [&](int i, float *xb) {
        float4 r = xr[i];
        if (XMK == 1)
            r = make_float4(xm[i].x > 0.f ? r.x : 0.f, xm[i].y > 0.f ? r.y : 0.f, xm[i].z > 0.f ? r.z : 0.f, xm[i].w > 0.f ? r.w : 0.f);
        if (XMK == 2) r = make_float4(r.x * (xm[i].x + m_one), r.y * (xm[i].y + m_one), r.z * (xm[i].z + m_one), r.w * (xm[i].w + m_one));
        st4(xdst[i] >= 0 ? xb + xdst[i] : scratch, r);
    };
    float bias_on = 1.f;                          // 0 for the clamped re-store of the last batch
    auto store_y = [&](int i, float *yb) {
        const int sl = tid + i * NT;
        float4 r = yr[i];
        if (GM) r = make_float4(ym[i].x > 0.f ? r.x : 0.f, ym[i].y > 0.f ? r.y : 0.f, ym[i].z > 0.f ? r.z : 0.f, ym[i].w > 0.f ? r.w : 0.f);
        st4(yb + (sl / YQ) * GW_CO + (sl % YQ) * 4, r);
        bsum = make_float4(bsum.x + bias_on * r.x, bsum.y + bias_on * r.y, bsum.z + bias_on * r.z, bsum.w + bias_on * r.w);
    };

    // ---- row `wave` of the two transforms.  B^T d B: rows (ra, rb) of the window, te = d[ra] + sb * d[rb];
    // A dy: ca * g[0][.] + cb * g[1][.]  ((1,0), (1,1), (1,-1), (0,-1))
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    // A dy rows: g0, g0 + g1, g0 - g1, -g1.  Wave 3 reads row 1 in the place of row 0 with coefficient 0 on the second operand and
    // its sign — like the -r1 of position 3 in every wave — is applied once, to the accumulators, after the loop: one FMA per value
    const float cb = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
    const int yr0 = wave == 3 ? YW : 0;                             // pixel row offset of the first operand
    // tile of MFMA step st and K index kk: TXB = 8: (row 0, column 2 st + kk); TXB = 2: (row st, column kk)
    const int xa_off = (ra * PW + 2 * kk) * 32 + l31, xb_off = (rb * PW + 2 * kk) * 32 + l31;      // + st * G::SX
    const int y_off = (2 * kk) * GW_CO + l31;                                                       // + st * G::SY pixels
    float da[4], db[4], g0[NF][2], g1[NF][2];    // raw operands of the tile pair being prepared
    float an[2][4], bn[2][NF][4];               // operand sets of tile pairs st & 1 = 0 / 1 (four pairs per batch: the parity carries over)
    auto fetch_x = [&](const float *xc, int st) {
#pragma unroll
        for (int c = 0; c < 4; ++c) da[c] = xc[xa_off + st * G::SX + c * 32], db[c] = xc[xb_off + st * G::SX + c * 32];
    };
    auto fetch_y = [&](const float *yc, int st, int f0, int f1) {
#pragma unroll
        for (int f = f0; f < f1; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                g0[f][c] = yc[y_off + (yr0 + st * G::SY + c) * GW_CO + f * 32], g1[f][c] = yc[y_off + (YW + st * G::SY + c) * GW_CO + f * 32];
    };
    auto finish_x = [&](int o) {
        const float t0 = da[0] + sb * db[0], t1 = da[1] + sb * db[1], t2 = da[2] + sb * db[2], t3 = da[3] + sb * db[3];
        an[o][0] = t0 - t2, an[o][1] = t1 + t2, an[o][2] = t2 - t1, an[o][3] = t1 - t3;
    };
    auto finish_y = [&](int o, int f) {
        const float r0 = g0[f][0] + cb * g1[f][0], r1 = g0[f][1] + cb * g1[f][1];
        bn[o][f][0] = r0, bn[o][f][1] = r0 + r1, bn[o][f][2] = r0 - r1, bn[o][f][3] = r1;       // (position 3 and wave 3: sign applied at the end)
    };

    const int step = gridDim.x;
    int batch = blockIdx.x;
    if (batch < q.nbatch) {
        const int last = batch + ((q.nbatch - 1 - batch) / step) * step;
        load_first(batch);
        load_raw(batch);
#pragma unroll
        for (int i = 0; i < NXS; ++i) store_x(i, Xp);
#pragma unroll
        for (int i = 0; i < NYS; ++i) store_y(i, Yp);
        load_raw(min(batch + step, last));
        __syncthreads();
        fetch_x(Xp, 0), fetch_y(Yp, 0, 0, NF);      // operands of the first tile pair (later batches: prepared under the previous one)
        finish_x(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) finish_y(0, f);
        int cur = 0;
        for (; batch <= last; batch += step, cur ^= 1) {
            bias_on = batch + step <= last ? 1.f : 0.f;
            const int b2 = min(batch + 2 * step, last);
            const float *xc = Xp + cur * GR_XP, *yc = Yp + cur * GR_YP;
            float *xn = Xp + (cur ^ 1) * GR_XP, *yn = Yp + (cur ^ 1) * GR_YP;
            // staging slices: raw strip of the next batch -> the other LDS buffer, then the loads of the batch after it
            // (NXS = 3 (2 x 16 strips) or 2 (8 x 4) input slots, NYS = NF gradient slots per thread; SL slices per tile pair)
            constexpr int SL = NF == 4 ? 4 : 3, S0 = 3, S1 = S0 + NYS, S2 = S1 + 1, S3 = S2 + 3, S4 = S3 + NYS;
            static_assert(S4 <= 4 * SL, "staging slices fit the gaps of a batch");
            auto stage = [&](int k) {
                if (k < S0) { if (k < NXS) store_x(k, xn); }
                else if (k < S1) store_y(k - S0, yn);
                else if (k == S1) load_begin(b2);
                else if (k < S3) { if (k - S2 < NXS) load_x(k - S2); }
                else if (k < S4) load_y(k - S3);
            };
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int o = st & 1, on = o ^ 1;       // operand set in use / being prepared (no register copies)
                // the operands of the next tile pair are fetched / finished in the gaps; for the last pair of a batch that is the
                // first pair of the NEXT batch, whose raw strip is complete in the other buffer since the barrier behind pair 2
                auto gap = [&](int gidx) {          // compile-time constant after unrolling
                    if (NF == 1) {
                        if (gidx == 0) fetch_x(st < 3 ? xc : xn, (st + 1) & 3), fetch_y(st < 3 ? yc : yn, (st + 1) & 3, 0, 1);
                        if (gidx == 1) stage(st * 3), stage(st * 3 + 1);
                        if (gidx == 2) finish_x(on);
                        if (gidx == 3) finish_y(on, 0), stage(st * 3 + 2);
                    } else if (NF == 2) {
                        if (gidx == 0) fetch_x(st < 3 ? xc : xn, (st + 1) & 3);
                        if (gidx == 1) fetch_y(st < 3 ? yc : yn, (st + 1) & 3, 0, 2);
                        if (gidx == 2) stage(st * 3);
                        if (gidx == 3) stage(st * 3 + 1);
                        if (gidx == 4) finish_x(on);
                        if (gidx == 5) finish_y(on, 0);
                        if (gidx == 6) finish_y(on, 1);
                        if (gidx == 7) stage(st * 3 + 2);
                    } else {
                        if (gidx == 0) fetch_x(st < 3 ? xc : xn, (st + 1) & 3);
                        if (gidx == 1) fetch_y(st < 3 ? yc : yn, (st + 1) & 3, 0, 2);
                        if (gidx == 2) fetch_y(st < 3 ? yc : yn, (st + 1) & 3, 2, 4);
                        if (gidx == 3) stage(st * 4);
                        if (gidx == 5) stage(st * 4 + 1);
                        if (gidx == 7) finish_x(on);
                        if (gidx == 8) finish_y(on, 0);
                        if (gidx == 9) finish_y(on, 1);
                        if (gidx == 10) stage(st * 4 + 2);
                        if (gidx == 11) finish_y(on, 2);
                        if (gidx == 12) finish_y(on, 3);
                        if (gidx == 14) stage(st * 4 + 3);
                    }
                };
#pragma unroll
                for (int pl = 0; pl < 4; ++pl)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        __builtin_amdgcn_sched_barrier(0);
                        acc[pl][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[o][pl], bn[o][f][pl], acc[pl][f], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        gap(pl * NF + f);
                    }
                // raw strip of the next batch (stored under pairs 0 and 1) visible; this batch's strip was last read by the
                // fetches of pair 3 above (in pair 2's gaps), so its buffer is free for the stores of the next iteration
                if (st == 2) __syncthreads();
            }
        }
    }

    // D[row = input channel][col = output channel] of position 4 * wave + pl -> ws[(pos*Cin + c)*Cout + n]
    // dw_slabs > 0: the tile split blockIdx.x owns slab blockIdx.x of the workspace and joins it by a plain read-modify-write — the
    // launches of a layer are serialised on their stream and no other workgroup of this launch touches the slab, so the sums are
    // formed in a fixed order (bit-reproducible gradients); ramnet_reduce_slabs adds the slabs up, in order, when the pass ends.
    // dw_slabs == 0 (callers that pass one [16][Cin][Cout] workspace): the splits meet by atomic adds.
    const int Cin = s.Cin;
    const bool slabs = p.dw_slabs > 0;
    float *dwb = p.dw + (slabs ? (size_t)blockIdx.x * 16 * Cin * p.Cout : 0);
    if (slabs) {
        // software-pipelined: the 16 old values of group g + 1 are requested before group g is stored (the compiler cannot prove that the
        // stores do not alias the next loads and would otherwise expose one memory round trip per group: 8 per workgroup)
        float old[2][16];
        auto grp_load = [&](int g, float (&o)[16]) {
            const int pl = g / NF, f = g % NF, n = n0 + f * 32 + l31;
            const float *col = dwb + (size_t)(4 * wave + pl) * Cin * p.Cout + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                o[r] = (c < Cin && n < p.Cout) ? col[(size_t)c * p.Cout] : 0.f;
            }
        };
        grp_load(0, old[0]);
#pragma unroll
        for (int g = 0; g < 4 * NF; ++g) {
            if (g + 1 < 4 * NF) grp_load(g + 1, old[(g + 1) & 1]);
            const int pl = g / NF, f = g % NF, n = n0 + f * 32 + l31;
            float *col = dwb + (size_t)(4 * wave + pl) * Cin * p.Cout + n;
            const bool neg = (wave == 3) != (pl == 3);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (c < Cin && n < p.Cout) col[(size_t)c * p.Cout] = old[g & 1][r] + (neg ? -acc[pl][f][r] : acc[pl][f][r]);
            }
        }
    } else {
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = n0 + f * 32 + l31;
                float *col = dwb + (size_t)(4 * wave + pl) * Cin * p.Cout + n;
                const bool neg = (wave == 3) != (pl == 3);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (c < Cin && n < p.Cout) atomicAdd(col + (size_t)c * p.Cout, neg ? -acc[pl][f][r] : acc[pl][f][r]);
                }
            }
    }
    if (p.dbias != nullptr && blockIdx.y == 0) {
        __syncthreads();
        float *red = smem;                        // [NT / YQ][GW_CO]
        st4(red + (tid / YQ) * GW_CO + (tid % YQ) * 4, bsum);
        __syncthreads();
        if (tid < GW_CO) {
            float t = 0.f;
            for (int g = 0; g < NT / YQ; ++g) t += red[g * GW_CO + tid];
            if (n0 + tid < p.Cout) {
                if (slabs) p.dbias[(size_t)blockIdx.x * p.Cout + n0 + tid] += t;      // (one workgroup per (slab, channel block): blockIdx.y == 0)
                else atomicAdd(p.dbias + n0 + tid, t);
            }
        }
    }
}

// slab 0 += slab 1 + slab 2 + ... in that order (n floats per slab), the others are zeroed for the next pass
__global__ void reduce_slabs_kernel(float *__restrict__ ws, int slabs, size_t n) {
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        float4 a = ld4(ws + i);
        for (int s = 1; s < slabs; ++s) {
            const float4 b = ld4(ws + (size_t)s * n + i);
            a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            st4(ws + (size_t)s * n + i, f4zero());
        }
        st4(ws + i, a);
    }
}

// ws [16][CinWs][CoutWs] (dU) -> grad OIHW [Cout][Cin][3][3] (+=): dg = G^T dU G
__global__ void unpack_wgrad_wino_kernel(const float *__restrict__ ws, float *__restrict__ g, int Cout, int Cin, int CinWs, int CoutWs,
                                         int n_off, size_t total) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin), n = (int)(i / Cin);
        float u[4][4];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) u[a][b] = ws[((size_t)(a * 4 + b) * CinWs + c) * CoutWs + n_off + n];
        for (int ka = 0; ka < 3; ++ka)
            for (int kb = 0; kb < 3; ++kb) {
                float sum = 0.f;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) sum += G[a][ka] * u[a][b] * G[b][kb];
                g[((size_t)n * Cin + c) * 9 + ka * 3 + kb] += sum;
            }
    }
}

int launch_wgrad_wino(const ramnet_wgrad_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 9 && d.stride == 1);
    RAMNET_CHECK_ARG(d.in_mode != RAMNET_IN_UP2X && d.in_mode != RAMNET_IN_UP2X_SKIP);
    RAMNET_CHECK_ARG(d.Ho == d.Hin && d.Wo == d.Win);
    if (d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) RAMNET_CHECK_ARG(d.C0 % 32 == 0);   // a workgroup's channels come from one tensor
    auto log2_exact = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    if (d.in_mode == RAMNET_IN_S2D) RAMNET_CHECK_ARG(d.C0 >= 32 && log2_exact(d.C0) > 0);                 // ... or one parity group
    int dymin = 127, dxmin = 127;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin;
    }
    for (int t = 0; t < 9; ++t) {   // the workspace rows follow the forward tap order (kh*3 + kw): require exactly that list
        RAMNET_CHECK_ARG(d.dy[t] - dymin == t / 3 && d.dx[t] - dxmin == t % 3);
        seen |= 1u << t;
    }
    RAMNET_CHECK_ARG(seen == 0x1ffu);
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    WgradWinoParams q;
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    if (d.in_mode == RAMNET_IN_S2D) q.src.Cin = 4 * d.C0, q.src.ld1 = log2_exact(d.C0);
    q.bx_n = cdiv(d.Wo, 16), q.ty_n = cdiv(d.Ho, 2);
    q.nbatch = q.bx_n * q.ty_n * d.B;
    q.dy0 = dymin, q.dx0 = dxmin;
    // batches of 8 tiles: a 2 x 16 or an 8 x 4 pixel strip, whichever pads the map less
    const bool tall = (long)cdiv(d.Wo, 4) * 4 * cdiv(d.Ho, 8) * 8 < (long)cdiv(d.Wo, 16) * 16 * cdiv(d.Ho, 2) * 2;
    if (tall) q.bx_n = cdiv(d.Wo, 4), q.ty_n = cdiv(d.Ho, 8), q.nbatch = q.bx_n * q.ty_n * d.B;
    // (A 128-channel workgroup — NF = 4: 256 accumulators, one workgroup per CU, 7.25 instead of 11 instructions per MFMA — was built in
    // round 3: six ConvGRU launches 1.522 -> 1.434 ms alone, training step 200.2 -> 190.5 samples/s co-scheduled; its launch path and
    // environment knob were removed in round 4, the kernel template keeps the parameter.)
    const int nf = g_opt_wgrad_wino_nf;
    const int gy = cdiv(q.src.Cin, 32), gz = cdiv(d.Cout, 32 * nf);
    // co-scheduled with the backward-data chain on another stream (the training step): 384 workgroups leave it room (measured
    // 256 ... 512: profiles/r03_h_tuning_notes.md)
    int splits = g_opt_wgrad_wino_blocks * (2 / nf) / (gy * gz);            // (<= WGRAD_WINO_TARGET: the slabs are sized for that)
    if (splits > q.nbatch) splits = q.nbatch;
    if (d.dw_slabs > 0 && splits > d.dw_slabs) splits = d.dw_slabs;
    if (splits < 1) splits = 1;
    const dim3 grid(splits, gy, gz);
    const size_t lds = ((size_t)2 * ((tall ? GrGeom<2>::XP : GrGeom<8>::XP) + 32 * 32 * nf) + 256 * 4) * sizeof(float);      // strips + scratch cells
    const int xmk = d.in_mode == RAMNET_IN_RELUMASK ? 1 : d.in_mode == RAMNET_IN_CAT_MUL ? 2 : 0;
    const bool gm = d.gmask != nullptr;
    {
        const unsigned long long px = (unsigned long long)d.Hin * d.Win, ldx = d.ld0 > d.ld1 ? d.ld0 : d.ld1;
        RAMNET_CHECK_ARG(px * (d.in_mode == RAMNET_IN_S2D ? 4 : 1) * ldx * 4ull < WOOB && px * d.ldm * 4ull < WOOB &&
                         (unsigned long long)d.Ho * d.Wo * d.ldg * 4ull < WOOB && (unsigned long long)d.Ho * d.Wo * d.ldgm * 4ull < WOOB);
    }
    note_kernel("conv_wgrad_wino_r_kernel<%d,%d,%d,%d>", xmk, (int)gm, tall ? 2 : 8, nf);
#define RAMNET_GO(XMv, GMv)                                                                                                  \
    do {                                                                                                                     \
    if (tall && nf == 1) hipLaunchKernelGGL((conv_wgrad_wino_r_kernel<XMv, GMv, 2, 1>), grid, dim3(256), lds, st, d, q); \
    else if (nf == 1) hipLaunchKernelGGL((conv_wgrad_wino_r_kernel<XMv, GMv, 8, 1>), grid, dim3(256), lds, st, d, q);    \
    else if (tall) hipLaunchKernelGGL((conv_wgrad_wino_r_kernel<XMv, GMv, 2, 2>), grid, dim3(256), lds, st, d, q);       \
    else hipLaunchKernelGGL((conv_wgrad_wino_r_kernel<XMv, GMv, 8, 2>), grid, dim3(256), lds, st, d, q);                 \
    } while (0)
    if (xmk == 1 && gm) RAMNET_GO(1, true);
    else if (xmk == 1) RAMNET_GO(1, false);
    else if (xmk == 2 && gm) RAMNET_GO(2, true);
    else if (xmk == 2) RAMNET_GO(2, false);
    else if (gm) RAMNET_GO(0, true);
    else RAMNET_GO(0, false);
#undef RAMNET_GO
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_wgrad_wino_slabs(int Cin, int Cout) {
    const int s = WGRAD_WINO_TARGET / (cdiv(Cin, 32) * cdiv(Cout, 64));
    return s < 1 ? 1 : s;
}

extern "C" int ramnet_reduce_slabs(float *ws, int slabs, size_t n, void *stream) {
    RAMNET_CHECK_ARG(ws && slabs >= 1 && n > 0 && n % 4 == 0 && ((uintptr_t)ws & 15) == 0);
    if (slabs == 1) return 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, slabs, n);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_unpack_wgrad_wino(const float *ws, float *grad, int Cout, int Cin, int CinWs, int CoutWs, int n_off, void *stream) {
    RAMNET_CHECK_ARG(ws && grad && Cout > 0 && Cin > 0 && CinWs >= Cin && n_off >= 0 && CoutWs >= n_off + Cout);
    const size_t total = (size_t)Cout * Cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(unpack_wgrad_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, grad, Cout, Cin, CinWs, CoutWs, n_off, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
