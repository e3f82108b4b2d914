// Winograd F(2x4, 3x3) backward-weights for gfx950 (MI355X): the 3x3 stride-1 layers of the RAM-Net path at the training batch (ConvGRU
// gates / candidate, residual blocks: submodules.py:447-452, 200-215), exact-fp32 arithmetic on v_mfma_f32_32x32x2_f32.
//
//   y = A_r^T [ (G_r g G_c^T) .* (B_r^T d B_c) ] A_c   (F(2,3) along rows, F(4,3) along columns: csrc/conv_wino6.hip)
//   =>  dU_p[ci][co] = sum_tiles V_p[tile][ci] * Z_p[tile][co],  V = B_r^T d B_c (4 x 6 window -> 4 x 6),  Z = A_r dy A_c^T (2 x 4 -> 4 x 6),
//       dg = G_r^T dU G_c (ramnet_unpack_wgrad_wino2x4): 24 multiplies per 8 outputs and channel pair instead of the 32 of F(2x2,3x3).
//
// Same formulation as csrc/conv_wgrad_wino.hip — the four waves own the four ROWS of the transform grid, one MFMA step reduces over two
// tiles, lane (channel, tile parity) forms row `wave` of both transforms in registers out of raw LDS strips, and those are the A / B
// operands — with six column positions per wave instead of four: a workgroup owns 32 input x 32 output channels (6 x 16 = 96 accumulators,
// two workgroups per CU: the shape the co-scheduled training step wants), a batch is 8 tiles = 64 output pixels (a 4 x 16,
// 8 x 8 or 16 x 4 pixel strip, whichever pads the map least), raw strips double-buffered with one barrier per batch, the tile splits of a
// launch own per-split slabs of a [S][24][Cin][Cout] workspace (plain read-modify-write: bit-reproducible).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

// Tuning builds only (tools/abl_wgrad6.sh): RAMNET_ABL is a bit mask that removes one ingredient of the main loop — 1 staging (global loads,
// LDS stores, batch walk), 2 LDS reads of the raw strips, 4 transform arithmetic, 8 barrier, 128 the whole loop, 256 the join.  Such a build
// computes WRONG results; only its duration means something.  The product is built with RAMNET_ABL = 0.
#ifndef RAMNET_ABL
#define RAMNET_ABL 0
#endif
#define RAMNET_OPQ(QQ) asm volatile("" : "+v"(QQ))

// Probe builds (tools/probe_wino6.sh, -DRAMNET_PROBE): thread 0 of every workgroup stamps the shader clock at the phase boundaries of its life
// (csrc/conv_wino6.hip has the forward kernel's); ramnet_probe_w6_read copies the table out.
#ifdef RAMNET_PROBE
namespace ramnet { __device__ unsigned long long g_probe_w6[16384 * 16]; }
#define RAMNET_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) ramnet::g_probe_w6[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define RAMNET_STAMP(k) do { } while (0)
#endif

namespace ramnet {

constexpr int WG6_TARGET = 384;      // workgroups per launch the tile splits aim at

constexpr int WG6_MAXSEG = RAMNET_WGRAD_MAX_SEGMENTS;

struct WgradWino6Params {
    InSrc src;
    int bx_n, ty_n, nbatch;     // strips per row, strip rows per image, total (all segments)
    int dy0, dx0;               // offset of the first filter tap
    int nseg, seg_images;       // segments of the launch (ramnet_wgrad_desc.segs; 1: the descriptor's own tensors), images per segment
    int splits, gy, gz, xcd_map; // tile splits, 32-channel input / output blocks; xcd_map: the 1-D grid deals the splits to the 8 XCDs
    ramnet_wgrad_seg seg[WG6_MAXSEG];
};

// TXB = tile columns per strip: 4 (4 x 16 output pixels), 2 (8 x 8) or 1 (16 x 4)
template <int TXB> struct G6Geom {
    static constexpr int TYB = 8 / TXB, YH = 2 * TYB, YW = 4 * TXB, PH = YH + 2, PW = YW + 2;
    static constexpr int XPIX = PH * PW, XSLOTS = XPIX * 8, NXS = (XSLOTS + 255) / 256;
    // LDS strips: [row][32 channels][2 buffers][row pitch] floats.  The MFMA operands put the lanes along the channels, so a lane's window row
    // (6 consecutive pixels) / gradient row (4) is contiguous: one ds_read_b128 + one ds_read_b64 / one ds_read_b128 at immediate offsets
    // (round 5: 20 -> 6 LDS reads per tile pair and wave).  RPX: row pitch of the input strip (a multiple of 4 >= PW: 16-byte aligned tile
    // origins).  The two buffers of the double-buffered strips sit side by side inside a channel's row, so that EVERY address of the loop is
    // a per-thread base + an immediate small enough for ds_write2_b32's 8-bit offsets — the four channels of a staged 16-byte slot lie
    // CH, 2 CH, 3 CH floats apart, the other buffer RPX / YW further: no address arithmetic in the loop at all.  CHX / CHY = channel pitch,
    // 2 x row pitch + 4 so that CH / 4 is ODD: the 16 lanes a ds_read_b128 serves per cycle fall on 16 distinct 4-bank groups.
    static constexpr int RPX = (PW + 3) / 4 * 4, CHX = 2 * RPX + 4, CHY = 2 * YW + 4;
    static constexpr int ROWX = 32 * CHX, ROWY = 32 * CHY;
    static constexpr int XTOT = PH * ROWX;         // raw input strips, both buffers
    static constexpr int YTOT = YH * ROWY;         // raw gradient strips, both buffers
    static_assert(RPX > PW && 3 * CHX + RPX < 256 && 3 * CHY + YW < 256, "pad pixel per row; ds_write2_b32 offsets");
    // tile pair st (tiles 2 st, 2 st + 1; the lane's tile = 2 st + kk): offsets of its window / output origin that do not depend on the lane
    static constexpr int sx(int st) { return TXB == 4 ? (st >> 1) * 2 * ROWX + (st & 1) * 8 : TXB == 2 ? st * 2 * ROWX : st * 4 * ROWX; }
    static constexpr int sy(int st) { return TXB == 4 ? (st >> 1) * 2 * ROWY + (st & 1) * 8 : TXB == 2 ? st * 2 * ROWY : st * 4 * ROWY; }
};

// XMK: second operand of the input loader — 0 none, 1 ReLU mask (x * (xm > 0)), 2 product (the h*r half of a CAT_MUL input); GM: ReLU mask on dy
template <int XMK, bool GM, int TXB>
__global__ void __launch_bounds__(256, 2) conv_wgrad_wino_r6_kernel(const ramnet_wgrad_desc p, const WgradWino6Params q) {
    using G = G6Geom<TXB>;
    constexpr int NT = 256, XQ = 8, YQ = 8, NXS = G::NXS, NYS = 2, XSLOTS = G::XSLOTS;
    constexpr int PW = G::PW, YW = G::YW, RPX = G::RPX, CHX = G::CHX, CHY = G::CHY, ROWX = G::ROWX, ROWY = G::ROWY;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xp = smem;                   // [PH][32][2][RPX] (+ 4 pad floats per channel)
    float *Yp = smem + G::XTOT;         // [YH][32][2][YW]  (+ 4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    RAMNET_STAMP(0);
#ifdef RAMNET_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_probe_w6[blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memrealtime();
        g_probe_w6[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_probe_w6[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    // Workgroup -> (tile split, input block, output block).  The gy x gz workgroups of one split read the SAME strips (every input block
    // the split's dy strips, every output block its x strips): consecutive workgroup ids go round-robin to the 8 XCDs (private L2s), so
    // with splits % 8 == 0 XCD x is dealt the splits x, x + 8, ... and the workgroups of one split sit in consecutive slots of ONE XCD —
    // a strip is fetched into one L2 instead of up to eight (L2 hit rate of the scale-0 launches 55-61 % before: profiles/r05_z_pmc_l2.txt).
    int sp_i, by_i, bz_i;
    if (q.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = q.gy * q.gz;
        sp_i = (slot / per) * 8 + xcd;
        const int rem = slot % per;
        by_i = rem % q.gy, bz_i = rem / q.gy;
    } else {
        sp_i = blockIdx.x % q.splits;
        const int rem = blockIdx.x / q.splits;
        by_i = rem % q.gy, bz_i = rem / q.gy;
    }
    const int c0 = by_i * 32, n0 = bz_i * 32;
    const InSrc &s = q.src;

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- raw-data prefetch (as in conv_wgrad_wino.hip: buffer loads at lane + scalar offsets, out-of-image = out-of-range offset)
    float4 xr[NXS], xm[NXS], yr[NYS], ym[NYS];
    const bool second = (s.mode == RAMNET_IN_CAT || s.mode == RAMNET_IN_CAT_MUL) && c0 >= s.C0;
    // space-to-depth view of a stride-2 5x5 encoder's input (RAMNET_IN_S2D, s.ld1 = log2 C0): the workgroup's 32 channels lie in ONE parity
    // group g = (a, b) — logical pixel (iy, ix) is stored pixel (2 iy + a, 2 ix + b) of the [2 Hin][2 Win][C0] image: doubled pixel strides,
    // (a, b) and the channel base folded into the buffer's base pointer
    const bool s2d = s.mode == RAMNET_IN_S2D;
    const int s2g = s2d ? c0 >> s.ld1 : 0, pxs = s2d ? 2 : 1, WinS = pxs * s.Win;
    const bool use_m = XMK == 1 || (XMK == 2 && second);
    const float m_one = use_m ? 0.f : 1.f;
    const int ldS = second ? s.ld1 : s.ld0;
    int xpy[NXS], xpx[NXS], ypx[NYS], ypy[NYS], xdst[NXS];
    unsigned xoff[NXS], xmoff[NXS], yoff[NYS], ymoff[NYS];
    bool xslot[NXS], yslot[NYS];
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
        const int sl = tid + i * NT, qd = sl % XQ, pix = sl / XQ;
        xpy[i] = pix / PW, xpx[i] = pix - xpy[i] * PW;
        xslot[i] = sl < XSLOTS && c0 + qd * 4 < s.Cin;
        xoff[i] = (unsigned)((pxs * xpy[i] * WinS + pxs * xpx[i]) * ldS + qd * 4) * 4u;
        xmoff[i] = (unsigned)((xpy[i] * s.Win + xpx[i]) * s.ldm + qd * 4) * 4u;
        // (threads without a slot write the pad pixel PW of row 0 of their four channels: never read, and the stores keep immediate offsets)
        xdst[i] = (qd * 4) * CHX + (sl < XSLOTS ? xpy[i] * ROWX + xpx[i] : PW);
    }
#pragma unroll
    for (int i = 0; i < NYS; ++i) {
        const int sl = tid + i * NT, qd = sl % YQ, pix = sl / YQ;
        ypx[i] = pix % YW, ypy[i] = pix / YW;
        yslot[i] = n0 + qd * 4 < p.Cout;
        yoff[i] = (unsigned)((ypy[i] * p.Wo + ypx[i]) * p.ldg + n0 + qd * 4) * 4u;
        ymoff[i] = (unsigned)((ypy[i] * p.Wo + ypx[i]) * p.ldgm + n0 + qd * 4) * 4u;
    }
    // A workgroup (tile split blockIdx.x) walks a CONTIGUOUS range of batches [lo, hi) of the launch's segments x images x strips, one
    // batch at a time: the walk is an increment with carries, and the segment — the tensors of one deferred ConvGRU cell update
    // (ramnet_wgrad_desc.segs) — changes at most a few times per workgroup: a real, almost never taken branch rebuilds the descriptors.
    int lb_ty = 0, lb_bx = 0, lb_b = 0, lb_seg = 0, lb_batch = 0;
    const int pad_m = -(pxs * q.dy0 * WinS + pxs * q.dx0);
    const int padS = pad_m > 0 ? pad_m : 0;
    unsigned so_x = 0, so_m = 0, so_g = 0, so_gm = 0;
    auto rx = wino_rsrc(nullptr, 0u);
    auto rmk = rx, rg = rx, rgm = rx;
    auto set_seg = [&](int sg) {
        const ramnet_wgrad_seg &g = q.seg[sg];
        const float *xsrc = second ? g.x1 + (c0 - s.C0) : s2d ? g.x0 + (c0 - (s2g << s.ld1)) + (long)((s2g >> 1) * WinS + (s2g & 1)) * ldS : g.x0 + c0;
        const float *msrc = s.mode == RAMNET_IN_RELUMASK ? g.xm + c0 : g.xm + (c0 - s.C0);
        rx = wino_rsrc(xsrc - (long)padS * ldS, WOOB);
        rmk = XMK ? wino_rsrc(msrc - (long)padS * s.ldm, WOOB) : rx;
        rg = wino_rsrc(g.dout, WOOB);
        rgm = GM ? wino_rsrc(g.gmask, WOOB) : rg;
    };
    // Byte offsets of the thread's slots with the ROW half of the in-image test folded in (WOOB: the slot's row of this strip row lies outside
    // the map, or the thread has no slot): they change with the strip row only — once per bx_n batches, a real, rarely taken branch — so a
    // batch tests the columns alone (one add, one compare, one select per slot instead of two of each and a mask combination).
    unsigned xov[NXS], xmv[XMK ? NXS : 1], yov[NYS], ymv[GM ? NYS : 1];
    auto rows_update = [&]() {
#pragma unroll
        for (int i = 0; i < NXS; ++i) {
            const bool ok = xslot[i] & ((unsigned)(G::YH * lb_ty + q.dy0 + xpy[i]) < (unsigned)s.Hin);
            xov[i] = ok ? xoff[i] : WOOB;
            if (XMK) xmv[i] = (ok & use_m) ? xmoff[i] : WOOB;
        }
#pragma unroll
        for (int i = 0; i < NYS; ++i) {
            const bool ok = yslot[i] & (G::YH * lb_ty + ypy[i] < p.Ho);
            yov[i] = ok ? yoff[i] : WOOB;
            if (GM) ymv[i] = ok ? ymoff[i] : WOOB;
        }
    };
    auto load_first = [&](int batch) {          // (once, in front of the loop: the integer divisions)
        int tt = batch;
        lb_bx = tt % q.bx_n;
        tt /= q.bx_n;
        lb_ty = tt % q.ty_n;
        tt /= q.ty_n;
        lb_b = tt % q.seg_images;
        lb_seg = tt / q.seg_images;
        lb_batch = batch;
        set_seg(lb_seg);
        rows_update();
    };
    auto load_begin = [&](int batch) {
        {                                       // batch == lb_batch (first call, clamped tail) or lb_batch + 1
            const int adv = batch != lb_batch ? 1 : 0;
            lb_bx += adv;
            const int cx = lb_bx >= q.bx_n ? 1 : 0;
            lb_bx -= cx * q.bx_n;
            lb_ty += cx;
            const int cy = lb_ty >= q.ty_n ? 1 : 0;
            lb_ty -= cy * q.ty_n;
            lb_b += cy;
            if (lb_b >= q.seg_images) {         // (uniform; the loads already in flight keep the descriptors they were issued with)
                lb_b = 0;
                set_seg(++lb_seg);
            }
            if (cx) rows_update();              // (uniform)
        }
        lb_batch = batch;
        const int pix = (lb_b * p.Ho + G::YH * lb_ty) * p.Wo + YW * lb_bx;
        const int corner = s2d ? ((lb_b * 2 * s.Hin + 2 * (G::YH * lb_ty + q.dy0)) * WinS + 2 * (YW * lb_bx + q.dx0))
                               : pix + q.dy0 * s.Win + q.dx0;
        so_x = (unsigned)((corner + padS) * ldS) * 4u;
        if (XMK) so_m = (unsigned)((corner + padS) * s.ldm) * 4u;
        so_g = (unsigned)(pix * p.ldg) * 4u;
        if (GM) so_gm = (unsigned)(pix * p.ldgm) * 4u;
    };
    auto bload = [](decltype(rx) r, unsigned vo, unsigned so) {
        // (the batch offset is wave-uniform; when the register allocator parks it in a VGPR — this kernel runs at the SGPR limit — the
        // compiler otherwise wraps every load in a waterfall loop: 9 instructions and an exec-mask round trip per load)
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, __builtin_amdgcn_readfirstlane((int)so), 0));
    };
    auto load_x = [&](int i) {              // (the row half of the in-image test lives in xov / xmv: rows_update)
        const bool ok = (unsigned)(YW * lb_bx + q.dx0 + xpx[i]) < (unsigned)s.Win;
        xr[i] = bload(rx, ok ? xov[i] : WOOB, so_x);
        if (XMK) xm[i] = bload(rmk, ok ? xmv[i] : WOOB, so_m);
    };
    auto load_y = [&](int i) {
        const bool ok = YW * lb_bx + ypx[i] < p.Wo;
        yr[i] = bload(rg, ok ? yov[i] : WOOB, so_g);
        if (GM) ym[i] = bload(rgm, ok ? ymv[i] : WOOB, so_gm);
    };
    auto load_raw = [&](int batch) {
        load_begin(batch);
#pragma unroll
        for (int i = 0; i < NXS; ++i) load_x(i);
#pragma unroll
        for (int i = 0; i < NYS; ++i) load_y(i);
    };
    float4 bsum = f4zero();                      // bias gradient partial of channel quad (tid % YQ)
    auto store_x = [&](int i, float *xb) {
        float4 r = xr[i];
        if (XMK == 1)
            r = make_float4(xm[i].x > 0.f ? r.x : 0.f, xm[i].y > 0.f ? r.y : 0.f, xm[i].z > 0.f ? r.z : 0.f, xm[i].w > 0.f ? r.w : 0.f);
        if (XMK == 2) r = make_float4(r.x * (xm[i].x + m_one), r.y * (xm[i].y + m_one), r.z * (xm[i].z + m_one), r.w * (xm[i].w + m_one));
        float *d = xb + xdst[i];
        d[0] = r.x, d[CHX] = r.y, d[2 * CHX] = r.z, d[3 * CHX] = r.w;
    };
    float bias_on = 1.f;                          // 0 for the clamped re-store of the last batch
    auto store_y = [&](int i, float *yb) {
        const int sl = tid + i * NT;
        float4 r = yr[i];
        if (GM) r = make_float4(ym[i].x > 0.f ? r.x : 0.f, ym[i].y > 0.f ? r.y : 0.f, ym[i].z > 0.f ? r.z : 0.f, ym[i].w > 0.f ? r.w : 0.f);
        float *d = yb + ((sl % YQ) * 4) * CHY + ((sl / YQ) / YW) * ROWY + (sl / YQ) % YW;
        d[0] = r.x, d[CHY] = r.y, d[2 * CHY] = r.z, d[3 * CHY] = r.w;
        bsum = make_float4(bsum.x + bias_on * r.x, bsum.y + bias_on * r.y, bsum.z + bias_on * r.z, bsum.w + bias_on * r.w);
    };

    // ---- row `wave` of the two transforms.  B_r^T d: rows (ra, rb) of the window, te = d[ra] + sb * d[rb]; A_r dy: g[0][.] + cb * g[1][.]
    // ((1,0), (1,1), (1,-1), (0,-1): wave 3 reads row 1 in the place of row 0 with cb = 0, its sign is applied to the accumulators at the end)
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const float cb = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
    const int yr0 = wave == 3 ? ROWY : 0;
    // the lane's tile of a pair: TXB >= 2: the next tile column (4 pixels), TXB = 1: the next tile row (2 pixel rows)
    const int kx = TXB == 1 ? kk * 2 * ROWX : kk * 4, ky = TXB == 1 ? kk * 2 * ROWY : kk * 4;
    const int xa_off = l31 * CHX + ra * ROWX + kx, xb_off = l31 * CHX + rb * ROWX + kx;
    const int y_off = l31 * CHY + ky;
    float da[6], db[6], g0[4], g1[4];             // raw operands of the tile pair being prepared
    float an[2][6], bn[2][6];                     // operand sets of tile pairs st & 1 = 0 / 1
    auto fetch_x = [&](const float *xc, int st) {
        if (RAMNET_ABL & 2) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { RAMNET_OPQ(da[c]); RAMNET_OPQ(db[c]); }
        } else {
            const float4 a4 = ld4(xc + xa_off + G::sx(st)), b4 = ld4(xc + xb_off + G::sx(st));
            const float2 a2 = *reinterpret_cast<const float2 *>(xc + xa_off + G::sx(st) + 4), b2 = *reinterpret_cast<const float2 *>(xc + xb_off + G::sx(st) + 4);
            da[0] = a4.x, da[1] = a4.y, da[2] = a4.z, da[3] = a4.w, da[4] = a2.x, da[5] = a2.y;
            db[0] = b4.x, db[1] = b4.y, db[2] = b4.z, db[3] = b4.w, db[4] = b2.x, db[5] = b2.y;
        }
    };
    auto fetch_y = [&](const float *yc, int st) {
        if (RAMNET_ABL & 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { RAMNET_OPQ(g0[c]); RAMNET_OPQ(g1[c]); }
        } else {
            const float4 a4 = ld4(yc + y_off + yr0 + G::sy(st)), b4 = ld4(yc + y_off + ROWY + G::sy(st));
            g0[0] = a4.x, g0[1] = a4.y, g0[2] = a4.z, g0[3] = a4.w;
            g1[0] = b4.x, g1[1] = b4.y, g1[2] = b4.z, g1[3] = b4.w;
        }
    };
    // B_c^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]   (conv_wino6.hip)
    float tt[6];
    auto finish_x0 = [&](int o) {
        if (RAMNET_ABL & 4) { RAMNET_OPQ(an[o][0]); RAMNET_OPQ(an[o][5]); return; }
#pragma unroll
        for (int c = 0; c < 6; ++c) tt[c] = da[c] + sb * db[c];
        an[o][0] = fmaf(4.f, tt[0], fmaf(-5.f, tt[2], tt[4]));
        an[o][5] = fmaf(4.f, tt[1], fmaf(-5.f, tt[3], tt[5]));
    };
    auto finish_x1 = [&](int o) {
        if (RAMNET_ABL & 4) { RAMNET_OPQ(an[o][1]); RAMNET_OPQ(an[o][2]); RAMNET_OPQ(an[o][3]); RAMNET_OPQ(an[o][4]); return; }
        const float sa = fmaf(-4.f, tt[2], tt[4]), sd = fmaf(-4.f, tt[1], tt[3]), ua = tt[4] - tt[2], ud = tt[3] - tt[1];
        an[o][1] = sa + sd, an[o][2] = sa - sd, an[o][3] = fmaf(2.f, ud, ua), an[o][4] = fmaf(-2.f, ud, ua);
    };
    // Z row = A_c w, A_c^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    auto finish_y = [&](int o) {
        if (RAMNET_ABL & 4) { for (int c = 0; c < 6; ++c) RAMNET_OPQ(bn[o][c]); return; }
        const float w0 = g0[0] + cb * g1[0], w1 = g0[1] + cb * g1[1], w2 = g0[2] + cb * g1[2], w3 = g0[3] + cb * g1[3];
        const float e = w0 + w2, f = w1 + w3, e4 = fmaf(4.f, w2, w0), f4 = fmaf(4.f, w3, w1);
        bn[o][0] = w0, bn[o][1] = e + f, bn[o][2] = e - f, bn[o][3] = fmaf(2.f, f4, e4), bn[o][4] = fmaf(-2.f, f4, e4), bn[o][5] = w3;
    };

    int batch = (int)((long long)q.nbatch * sp_i / q.splits);
    const int last = (int)((long long)q.nbatch * (sp_i + 1) / q.splits) - 1;
    RAMNET_STAMP(1);
    if (batch <= last) {
        load_first(batch);
        load_raw(batch);
#pragma unroll
        for (int i = 0; i < NXS; ++i) store_x(i, Xp);
#pragma unroll
        for (int i = 0; i < NYS; ++i) store_y(i, Yp);
        load_raw(min(batch + 1, last));
        __syncthreads();
        RAMNET_STAMP(2);
        fetch_x(Xp, 0), fetch_y(Yp, 0);
        finish_x0(0), finish_x1(0), finish_y(0);
        RAMNET_STAMP(3);
        // One batch; CUR = the LDS buffer holding it, a compile-time constant (the loop below alternates two instantiations): every strip
        // address is then the thread's base + an immediate — with a run-time buffer index the loop carried 15 address additions per batch.
        auto iter = [&](auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            bias_on = batch + 1 <= last ? 1.f : 0.f;
            const int b2 = min(batch + 2, last);
            const float *xc = Xp + cur * RPX, *yc = Yp + cur * YW;                  // (the buffers interleave inside a channel's row)
            float *xn = Xp + (cur ^ 1) * RPX, *yn = Yp + (cur ^ 1) * YW;
            // staging slices: the raw strips of the next batch (in registers) -> the other LDS buffer (k = 0..5, all in front of the barrier
            // behind tile pair 2), then the loads of the batch after it (k = 6..12)
            auto stage = [&](int k) {
                if (RAMNET_ABL & 1) return;
                if (k < 4) { if (k < NXS) store_x(k, xn); }
                else if (k < 6) store_y(k - 4, yn);
                else if (k == 6) load_begin(b2);
                else if (k < 11) { if (k - 7 < NXS) load_x(k - 7); }
                else if (k < 13) load_y(k - 11);
            };
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int o = st & 1, on = o ^ 1;       // operand set in use / being prepared
                // the next tile pair is fetched / finished in the gaps; for the last pair of a batch that is the first pair of the NEXT batch,
                // whose raw strips are complete in the other buffer since the barrier behind pair 2
                auto gap = [&](int g) {
                    if (g == 0) fetch_x(st < 3 ? xc : xn, (st + 1) & 3);
                    if (g == 1) fetch_y(st < 3 ? yc : yn, (st + 1) & 3), stage(st * 3);
                    if (g == 2) finish_x0(on);
                    if (g == 3) finish_x1(on), stage(st * 3 + 1);
                    if (g == 4) finish_y(on);
                    if (RAMNET_ABL & 1024) {      // (timing only: the vector work of the transforms once more — what a split of the operands would add)
                        if (g == 3) { for (int c = 0; c < 6; ++c) RAMNET_OPQ(da[c]); finish_x0(on), finish_x1(on); }
                        if (g == 5) { for (int c = 0; c < 4; ++c) RAMNET_OPQ(g0[c]); finish_y(on); }
                    }
                    if (g == 5) { stage(st * 3 + 2); if (st == 3) stage(12); }
                };
#pragma unroll
                for (int pl = 0; pl < 6; ++pl) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(RAMNET_ABL & 512) || pl < 2) acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[o][pl], bn[o][pl], acc[pl], 0, 0, 0);
                    else asm volatile("" : : "v"(an[o][pl]), "v"(bn[o][pl]));
                    __builtin_amdgcn_sched_barrier(0);
                    gap(pl);
                }
                if (st == 2 && !(RAMNET_ABL & 8)) __syncthreads();
            }
        };
        if (!(RAMNET_ABL & 128)) {
            do {                                        // (batch <= last here; ONE exit, at the bottom: a second path into the join makes the
                iter(std::integral_constant<int, 0>{}); //  allocator copy the 96 accumulators — 256 VGPRs and 144-392 bytes of scratch)
                ++batch;
                if (batch <= last) {                    // (uniform)
                    iter(std::integral_constant<int, 1>{});
                    ++batch;
                }
            } while (batch <= last);
        }
    }

    RAMNET_STAMP(4);
    // D of position 6 * wave + pl -> the BLOCKED workspace [24 positions][Cin / 32][Cout / 32][64 lanes][16 accumulator registers]: a lane's
    // 16 values (rows c = (r & 3) + 8 (r >> 2) + 4 kk of column n = lane & 31) are 64 contiguous bytes, a wave's block 4 KB — the join of
    // a split's partial sums with its slab is 4 + 4 16-byte accesses per position instead of 16 + 16 4-byte ones at a Cout-float stride
    // (the join is ~12 % of a launch at the training batch: every workgroup runs it at the same time, behind the last MFMA).  Channels
    // beyond Cin / Cout inside a block hold zeros (their strips load as zeros).  ramnet_unpack_wgrad_wino2x4 reads the same layout.
    const int nCiB = q.gy, nCoB = q.gz;
    const bool slabs = p.dw_slabs > 0;
    const size_t slab_floats = (size_t)24 * nCiB * nCoB * 1024;
    float *dwb = p.dw + (slabs ? (size_t)sp_i * slab_floats : 0) + ((size_t)by_i * nCoB + bz_i) * 1024 + lane * 16;
    const size_t pos_stride = (size_t)nCiB * nCoB * 1024;
    const bool neg = wave == 3;
    if (RAMNET_ABL & 256) {
        if (acc[0][0] == 123.456f) dwb[0] = acc[1][0] + acc[2][0] + acc[3][0] + acc[4][0] + acc[5][0];
    } else if (slabs) {                // pipelined read-modify-write of the split's own slab: position pl + 1 is requested before pl is stored
        float4 old[2][4];
        auto grp_load = [&](int pl, float4 (&o)[4]) {
            const float *b = dwb + (size_t)(6 * wave + pl) * pos_stride;
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = ld4(b + 4 * v);
        };
        grp_load(0, old[0]);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
            if (pl + 1 < 6) grp_load(pl + 1, old[(pl + 1) & 1]);
            float *b = dwb + (size_t)(6 * wave + pl) * pos_stride;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 o = old[pl & 1][v];
                const float sg = neg ? -1.f : 1.f;
                st4(b + 4 * v, make_float4(fmaf(sg, acc[pl][4 * v], o.x), fmaf(sg, acc[pl][4 * v + 1], o.y), fmaf(sg, acc[pl][4 * v + 2], o.z),
                                            fmaf(sg, acc[pl][4 * v + 3], o.w)));
            }
        }
    } else {
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
            float *b = dwb + (size_t)(6 * wave + pl) * pos_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(b + r, neg ? -acc[pl][r] : acc[pl][r]);
        }
    }
#ifdef RAMNET_PROBE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RAMNET_STAMP(5);
    if (p.dbias != nullptr && by_i == 0) {
        __syncthreads();
        float *red = smem;                        // [NT / YQ][32]
        st4(red + (tid / YQ) * 32 + (tid % YQ) * 4, bsum);
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
            for (int g = 0; g < NT / YQ; ++g) t += red[g * 32 + tid];
            if (n0 + tid < p.Cout) {
                if (slabs) p.dbias[(size_t)sp_i * p.Cout + n0 + tid] += t;
                else atomicAdd(p.dbias + n0 + tid, t);
            }
        }
    }
#ifdef RAMNET_PROBE
    __builtin_amdgcn_s_waitcnt(0);
    RAMNET_STAMP(6);
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_probe_w6[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

// blocked ws [24][CinWs / 32][CoutWs / 32][64][16] (dU, position = 6 * row + column; layout: the kernel's join above) -> grad OIHW
// [Cout][Cin][3][3] (+=): dg = G_r^T dU G_c
__global__ void unpack_wgrad_wino2x4_kernel(const float *__restrict__ ws, float *__restrict__ g, int Cout, int Cin, int CinWs, int CoutWs,
                                            int n_off, size_t total) {
    const float G2[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const double G4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nCiB = (CinWs + 31) / 32, nCoB = (CoutWs + 31) / 32;
    const size_t pos_stride = (size_t)nCiB * nCoB * 1024;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin), n = (int)(i / Cin);
        const int nn = n_off + n, c32 = c & 31;
        const size_t at = ((size_t)(c >> 5) * nCoB + (nn >> 5)) * 1024 + ((nn & 31) + 32 * ((c32 >> 2) & 1)) * 16 + (c32 & 3) + 4 * (c32 >> 3);
        double u[4][6];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 6; ++b) u[a][b] = ws[(size_t)(a * 6 + b) * pos_stride + at];
        for (int ka = 0; ka < 3; ++ka)
            for (int kb = 0; kb < 3; ++kb) {
                double sum = 0;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 6; ++b) sum += G2[a][ka] * u[a][b] * G4[b][kb];
                g[((size_t)n * Cin + c) * 9 + ka * 3 + kb] += (float)sum;
            }
    }
}

bool wgrad_wino6_eligible(const ramnet_wgrad_desc &d) {
    return d.ntaps == 9 && d.stride == 1 && d.Ho == d.Hin && d.Wo == d.Win &&
           (d.in_mode == RAMNET_IN_PLAIN || d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK ||
            (d.in_mode == RAMNET_IN_S2D && d.C0 >= 32 && (d.C0 & (d.C0 - 1)) == 0));
}

int launch_wgrad_wino6(const ramnet_wgrad_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(wgrad_wino6_eligible(d));
    if (d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) RAMNET_CHECK_ARG(d.C0 % 32 == 0);   // a workgroup's channels come from one tensor
    int dymin = 127, dxmin = 127;
    for (int t = 0; t < 9; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin;
    }
    for (int t = 0; t < 9; ++t) RAMNET_CHECK_ARG(d.dy[t] - dymin == t / 3 && d.dx[t] - dxmin == t % 3);      // the forward tap order kh*3 + kw
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    WgradWino6Params q;
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    if (d.in_mode == RAMNET_IN_S2D) {       // Cin = the four parity groups; ld1 carries log2 C0
        int sh = 0;
        while ((1 << sh) < d.C0) ++sh;
        q.src.Cin = 4 * d.C0, q.src.ld1 = sh;
    }
    q.dy0 = dymin, q.dx0 = dxmin;
    // strips of 8 tiles: 4 x 16, 8 x 8 or 16 x 4 output pixels, whichever pads the map least (ties: the widest)
    long best = -1;
    int txb = 4;
    for (int t = 4; t >= 1; t >>= 1) {
        const long a = (long)cdiv(d.Wo, 4 * t) * 4 * t * cdiv(d.Ho, 16 / t) * (16 / t);
        if (best < 0 || a < best) best = a, txb = t;
    }
    // segments: the tensors of several launches of the same layer (deferred ConvGRU cell updates) walked as one batch dimension
    q.nseg = d.nseg > 0 ? d.nseg : 1, q.seg_images = d.B;
    RAMNET_CHECK_ARG(q.nseg <= WG6_MAXSEG && (d.nseg == 0 || d.segs != nullptr));
    for (int i = 0; i < q.nseg; ++i) {
        if (d.nseg > 0) q.seg[i] = d.segs[i];
        else q.seg[i].x0 = d.x0, q.seg[i].x1 = d.x1, q.seg[i].xm = d.xm, q.seg[i].dout = d.dout, q.seg[i].gmask = d.gmask;
        RAMNET_CHECK_ARG(q.seg[i].x0 && q.seg[i].dout && (!cat || q.seg[i].x1) && (d.xm == nullptr) == (q.seg[i].xm == nullptr) &&
                         (d.gmask == nullptr) == (q.seg[i].gmask == nullptr));
    }
    q.bx_n = cdiv(d.Wo, 4 * txb), q.ty_n = cdiv(d.Ho, 16 / txb), q.nbatch = q.bx_n * q.ty_n * d.B * q.nseg;
    const int gy = cdiv(q.src.Cin, 32), gz = cdiv(d.Cout, 32);
    int splits = g_opt_wgrad_wino_blocks / (gy * gz);
    if (splits > q.nbatch) splits = q.nbatch;
    if (d.dw_slabs > 0 && splits > d.dw_slabs) splits = d.dw_slabs;
    if (splits < 1) splits = 1;
    // splits dealt to the XCDs (kernel): from 8 splits up, a multiple of 8 (the nearest the slabs and the batches allow)
    q.xcd_map = 0;
#ifndef RAMNET_NO_XCD_MAP
    if (splits >= 8) {
        int s8 = (splits + 4) / 8 * 8;
        if (s8 > q.nbatch || (d.dw_slabs > 0 && s8 > d.dw_slabs)) s8 = splits / 8 * 8;
        splits = s8, q.xcd_map = 1;
    }
#endif
    q.splits = splits, q.gy = gy, q.gz = gz;
    const dim3 grid(splits * gy * gz);
    const int strips = txb == 4 ? G6Geom<4>::XTOT + G6Geom<4>::YTOT : txb == 2 ? G6Geom<2>::XTOT + G6Geom<2>::YTOT : G6Geom<1>::XTOT + G6Geom<1>::YTOT;
    const size_t lds = (size_t)(strips > 1024 ? strips : 1024) * sizeof(float);         // (the bias reduction reuses 32 x 32 floats)
    const int xmk = d.in_mode == RAMNET_IN_RELUMASK ? 1 : d.in_mode == RAMNET_IN_CAT_MUL ? 2 : 0;
    const bool gm = d.gmask != nullptr;
    {
        const unsigned long long px = (unsigned long long)d.Hin * d.Win * (d.in_mode == RAMNET_IN_S2D ? 4 : 1), ldx = d.ld0 > d.ld1 ? d.ld0 : d.ld1;
        RAMNET_CHECK_ARG(d.B * px * ldx * 4ull < WOOB && d.B * px * d.ldm * 4ull < WOOB &&
                         (unsigned long long)d.B * d.Ho * d.Wo * d.ldg * 4ull < WOOB && (unsigned long long)d.B * d.Ho * d.Wo * d.ldgm * 4ull < WOOB);
    }
    note_kernel("conv_wgrad_wino_r6_kernel<%d,%d,%d>", xmk, (int)gm, txb);
#define RAMNET_GO(XMv, GMv)                                                                                               \
    do {                                                                                                                  \
        if (txb == 4) hipLaunchKernelGGL((conv_wgrad_wino_r6_kernel<XMv, GMv, 4>), grid, dim3(256), lds, st, d, q);       \
        else if (txb == 2) hipLaunchKernelGGL((conv_wgrad_wino_r6_kernel<XMv, GMv, 2>), grid, dim3(256), lds, st, d, q);  \
        else hipLaunchKernelGGL((conv_wgrad_wino_r6_kernel<XMv, GMv, 1>), grid, dim3(256), lds, st, d, q);                \
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

#ifdef RAMNET_PROBE
extern "C" int ramnet_probe_w6_read(unsigned long long *dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ramnet::g_probe_w6), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int ramnet_wgrad_wino2x4_slabs(int Cin, int Cout) {
    const int s = WG6_TARGET / (cdiv(Cin, 32) * cdiv(Cout, 32));
    return s < 1 ? 1 : s;
}

extern "C" size_t ramnet_wgrad_wino2x4_ws_floats(int Cin, int Cout) { return (size_t)24 * cdiv(Cin, 32) * cdiv(Cout, 32) * 1024; }

extern "C" int ramnet_unpack_wgrad_wino2x4(const float *ws, float *grad, int Cout, int Cin, int CinWs, int CoutWs, int n_off, void *stream) {
    RAMNET_CHECK_ARG(ws && grad && Cout > 0 && Cin > 0 && CinWs >= Cin && n_off >= 0 && CoutWs >= n_off + Cout);
    const size_t total = (size_t)Cout * Cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(unpack_wgrad_wino2x4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, grad, Cout, Cin, CinWs, CoutWs, n_off, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
