// Winograd F(2x4, 3x3) with SPLIT OPERANDS for gfx950 (MI355X): the forward / backward-data launches of conv_wino6.hip (ConvGRU gates and
// candidate, submodules.py:447-452; residual blocks, :200-215; the stride-2 5x5 encoders over their space-to-depth view) with every
// product of the Winograd domain evaluated on the bf16 matrix pipe at fp32 accuracy:
//
//   x = x1 + x2 + x3   (three bf16 terms, each the round-to-nearest bf16 of what the terms before it left: 24+ significand bits, the
//                       residuals x - x1 and x - x1 - x2 are exact in fp32)
//   a . b  ~=  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1          (the dropped terms a2 b3 + a3 b2 + a3 b3 are <= 2^-25 |a b|)
//
// six v_mfma_f32_32x32x16_bf16 (fp32 accumulation; 32 cycles for 16 channels) instead of eight v_mfma_f32_32x32x2_f32 (64 cycles for 2
// channels each): 192 against 512 matrix-pipe cycles per 16 input channels.  The weights are split ONCE when they are packed (three bf16
// planes in B-operand lane order, 6 bytes per weight); the transformed activations are split in registers where they are built.
//
// What bounds the loop is no longer the matrix pipe but what feeds it, and the shape of the workgroup follows from that budget
// (profiles/r06_split_design.md):
//   * building one A operand (one position x 32 tiles x 16 channels: row combination, column transform B4, two residual levels) costs
//     ~70 vector instructions per lane against 6 x 32 cycles of MFMAs per 32 output channels: each wave therefore multiplies every A
//     operand with TWO 32-channel weight blocks (64 output channels per workgroup, 192 accumulators): 6.7 other instructions per MFMA;
//   * 192 accumulators + the operand rings need one wave per SIMD (512 registers): ONE 256-thread workgroup per CU; a wave interleaves
//     its own vector work with its own MFMAs (bf16 MFMAs leave the issue port free, MI355X_MICROARCH.md);
//   * the B operands stream from L2 at 0.5 KB per MFMA (3 planes x 1 KB per 6 MFMAs): ~45 B/clk per CU at the target rate, through a
//     ring of four positions in registers (requested three positions ahead), ONE memory instruction per MFMA gap: the four waves run in
//     step and share the CU's address unit — bursts of six loads cost 17 % of a launch.
// Measured (profiles/r06_split_design.md, r06_probe_split_after.txt, r06_pmc_split_vs_exact.txt): x1.55 on the six forward launches of a
// ConvGRU update, 3630-3990 cycles per 16-channel chunk for 2304 cycles of MFMAs (issue-bound: the wave issues 45-67 % of its cycles, a
// lone wave adds its vector and memory instructions to its MFMA time), 10-15 us of a workgroup's 26-76 outside the main loop.
// Everything around the main loop follows conv_wino6.hip: wave w owns ROW w of the 4 x 6 transform grid, only the raw patch goes
// through LDS (skewed conflict-free layout, fused loaders), the waves meet once in LDS (both 32-channel halves: 136 KB) for the output
// transform and the straight-line channel-quad epilogue.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.hpp"
#include "conv_epilogue.hpp"
#include "conv_wino_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace ramnet {

// Tuning builds only (tools/abl_wino6s.sh, -DRAMNET_ABL6S=<mask>): parts of the main loop removed to see what each costs (results are WRONG):
// 1 patch global loads, 2 B-operand loads, 4 split, 8 column transform, 16 next row (LDS reads + combinations), 32 MFMAs, 64 patch LDS stores,
// 256 the MFMAs of a position alternate between four accumulators instead of two.
// Such builds only instantiate the concatenation loader on the 16 x 16 and 32 x 8 tiles.
#ifndef RAMNET_ABL6S
#define RAMNET_ABL6S 0
#endif
// Probe builds (tools/probe_wino6.sh, -DRAMNET_PROBE): thread 0 of every workgroup stamps the shader clock at the phase boundaries of its life into
// g_probe6s[workgroup][16] (slot 7 / 10: the 100 MHz wall counter at entry / exit, 8 / 9: HW_ID / XCC_ID); ramnet_probe6s_read copies the table out.
#ifdef RAMNET_PROBE
__device__ unsigned long long g_probe6s[16384 * 16];
#define W6S_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_probe6s[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W6S_STAMP(k) do { } while (0)
#endif
#define ABL6S(bit) ((RAMNET_ABL6S & (bit)) != 0)
#define ABL6S_KEEP(x) asm volatile("" : : "v"(x))

constexpr int WKS = 16;                              // input channels per chunk = K of one v_mfma_f32_32x32x16_bf16
constexpr int W6S_BN = 64;                           // output channels per workgroup
constexpr int W6S_POS_BYTES = 2 * 3 * 1024;          // B operands of one (row, position): [32-channel half 2][plane 3][lane 64][8 bf16]
constexpr int W6S_BLK_BYTES = 4 * 6 * W6S_POS_BYTES; // one (chunk, 64-channel block): 144 KB

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): every index a compile-time constant (register arrays, op tables)
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// Patch geometry as in conv_wino6.hip (R6Geom) with FOUR channel-quad planes per buffer.  Planes are padded to 2 mod 8 slots: the four
// quads of a pixel are stored by four neighbouring lanes (ds_write_b128 serves 8 lanes per cycle: two pixels x four planes -> 8 distinct
// bank groups); a read touches one plane per lane group, where the row skew keeps the 16 lanes apart.
template <int TXG> struct R6SGeom {
    static constexpr int TYG = 32 / TXG, TH = 2 * TYG, TW = 4 * TXG, PH = TH + 2, PW = TW + 2, PWS = PW + 3;
    static constexpr int PSLOTS = PH * PWS + ((2 - (PH * PWS) % 8) + 8) % 8;
    static constexpr int PLANE = PSLOTS * 4;         // floats of one channel-quad plane of the patch
    static constexpr int PFLOATS = 4 * PLANE;        // one buffer: 16 channels
    static_assert(PH * PW * 4 <= 1536, "six patch slots per thread");
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // {bf16(lo), bf16(hi)}, round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// the six partial products in issue order: (term of A, term of B), smallest first
__device__ constexpr int prod_a(int i) { return i == 0 ? 2 : i == 1 ? 1 : i == 2 ? 0 : i == 3 ? 1 : 0; }
__device__ constexpr int prod_b(int i) { return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : i == 3 ? 0 : i == 4 ? 1 : 0; }
// vector instructions of the column transform B4 of position q for 8 channels (positions (1, 2) and (3, 4) share sub-expressions)
__device__ constexpr int colops_n(int q) { return (q == 0 || q == 5) ? 16 : (q == 1 || q == 3) ? 24 : 8; }

// What a wave does BEHIND the twelve MFMAs of position P.  Memory instructions sit at fixed gaps, ONE per gap — the four waves of the workgroup run in
// step (a barrier per chunk) and share the CU's address unit, which takes 16 cycles per 1 KB wave-load: a burst of six loads per wave holds it for
// 384 cycles and stalls the issuing waves in front of their next MFMA (measured: B loads in one burst cost 17 % of the launch) —: gap 0, 2, .. 10
// the six B loads of the position three ahead, gap 1 the patch store, gap 3 the patch reload.  The vector / LDS work is a list of segments
// executed in order and cut into twelve equal slices.
// The row t = d[ra] + sb d[rb] lives in ONE register set: the next chunk's column j replaces this chunk's as soon as its last reader — the
// column transform that builds the A operands of position Q = P + 1 — is through with it: column 0 is only read by position 0 (built behind
// position 5 of the previous chunk), columns 2 and 4 last by position 3 (built behind 2), columns 1, 3, 5 last by position 5 (built behind 4).
enum { SEG_B = 0, SEG_PST, SEG_PLD, SEG_RD, SEG_COL, SEG_SPLIT, SEG_CMB };
struct Seg { int kind, col, slot; };                 // SEG_RD / SEG_CMB: column of the next row, raw register slot
struct OpRef { int kind, col, slot, idx; };
__device__ constexpr int seg_count(int P) { return (P == 0 ? 5 : P == 1 ? 7 : P == 2 ? 5 : P == 3 ? 9 : P == 4 ? 9 : 7) - 3; }
__device__ constexpr Seg seg_at(int P, int i0) {
    const int i = i0 + 3;
    if (P == 0 || P == 2) return i == 3 ? Seg{SEG_COL, 0, 0} : Seg{SEG_SPLIT, 0, 0};
    if (P == 1) return i == 3 ? Seg{SEG_RD, 0, 0} : i == 4 ? Seg{SEG_COL, 0, 0} : i == 5 ? Seg{SEG_SPLIT, 0, 0} : Seg{SEG_CMB, 0, 0};
    if (P == 3) return i == 3 ? Seg{SEG_RD, 2, 0} : i == 4 ? Seg{SEG_RD, 4, 1} : i == 5 ? Seg{SEG_COL, 0, 0} : i == 6 ? Seg{SEG_SPLIT, 0, 0}
                     : i == 7 ? Seg{SEG_CMB, 2, 0} : Seg{SEG_CMB, 4, 1};
    if (P == 4) return i == 3 ? Seg{SEG_COL, 0, 0} : i == 4 ? Seg{SEG_RD, 5, 0} : i == 5 ? Seg{SEG_RD, 1, 1} : i == 6 ? Seg{SEG_SPLIT, 0, 0}
                     : i == 7 ? Seg{SEG_CMB, 5, 0} : Seg{SEG_CMB, 1, 1};
    return i == 3 ? Seg{SEG_RD, 3, 0} : i == 4 ? Seg{SEG_COL, 0, 0} : i == 5 ? Seg{SEG_SPLIT, 0, 0} : Seg{SEG_CMB, 3, 0};
}
__device__ constexpr int seg_len(int P, Seg s) {
    return s.kind == SEG_B ? 6 : s.kind == SEG_PST ? 1 : s.kind == SEG_PLD ? 1 : s.kind == SEG_RD ? 4 : s.kind == SEG_COL ? colops_n((P + 1) % 6)
         : s.kind == SEG_SPLIT ? 44 : 8;
}
__device__ constexpr int ops_total(int P) {
    int n = 0;
    for (int i = 0; i < seg_count(P); ++i) n += seg_len(P, seg_at(P, i));
    return n;
}
__device__ constexpr OpRef op_at(int P, int k) {
    for (int i = 0; i < seg_count(P); ++i) {
        const Seg s = seg_at(P, i);
        const int n = seg_len(P, s);
        if (k < n) return {s.kind, s.col, s.slot, k};
        k -= n;
    }
    return {-1, 0, 0, 0};
}

template <int TXG, int MODE>
__global__ void __launch_bounds__(256, 1) conv_wino_r6s_kernel(const ramnet_conv_desc p, const WinoParams q) {
    constexpr int RO_LD = 64 + 4;                    // row of the exchange buffer [wave 4][column 4][tile 32][64 channels + pad]: both halves at once
    using G = R6SGeom<TXG>;
    constexpr int PL = G::PLANE, PF = G::PFLOATS, RPW = G::PWS, RTW = G::TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                             // [2 buffers][4 quads][PH x PW pixels][4] + 256 scratch cells; the epilogue reuses the space

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hq = lane >> 5;
    W6S_STAMP(0);
#ifdef RAMNET_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_probe6s[blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memrealtime();
        g_probe6s[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        g_probe6s[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    }
#endif

    // XCD-aware order as in conv_wino6.hip: the 64-channel blocks of ONE spatial tile are consecutive on one XCD
    // (quotients through the float reciprocals the launcher passes — the operands are far below 2^23, one correction step makes them exact —
    // instead of three integer divisions: ~0.5 us of a workgroup's set-up, which nothing else on this CU covers)
    auto fdiv = [](int n, int d, float inv) {
        int qv = (int)((float)n * inv);
        const int r = n - qv * d;
        qv += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
        return qv;
    };
    const int xslot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int nbl = q.nblk >> q.xg;
    const int xq = fdiv(xslot, nbl, q.inv_nbl);
    const int nblk_i = ((xslot - xq * nbl) << q.xg) + (xcd & ((1 << q.xg) - 1));
    int bid = xq * (8 >> q.xg) + (xcd >> q.xg);
    if (bid >= q.tiles_x * q.tiles_y * p.B) return;
    const int bq = fdiv(bid, q.tiles_x, q.inv_tx);
    const int tx_i = bid - bq * q.tiles_x;
    const int b = fdiv(bq, q.tiles_y, q.inv_ty);
    const int ty_i = bq - b * q.tiles_y;
    const int n0 = nblk_i * W6S_BN;
    const int oy0 = ty_i * G::TH, ox0 = tx_i * G::TW;
    const int iy0 = oy0 + q.dy0, ix0 = ox0 + q.dx0;

    // row `wave` of B2^T d: rows (ra, rb) of the tile's 4 x 6 window, t[j] = d[ra][j] + sb * d[rb][j]; the lane holds channels 8 hq .. 8 hq + 7
    // of tile l31 (quads 2 hq and 2 hq + 1): that IS the A operand layout of the 32x32x16 MFMA (row = lane & 31, k = 8 (lane >> 5) + 0..7)
    const int tty = l31 / TXG, ttx = l31 % TXG;
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int pra = 2 * hq * PL + ((2 * tty + ra) * RPW + 4 * ttx + (((2 * tty + ra) >> 1) & 3)) * 4;
    const int prb = 2 * hq * PL + ((2 * tty + rb) * RPW + 4 * ttx + (((2 * tty + rb) >> 1) & 3)) * 4;

    f32x16 acc[6][2];
    sfor<6>([&](auto I) { sfor<2>([&](auto Fh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[decltype(I)::value][decltype(Fh)::value][r] = 0.f;
    }); });

    const int nch = q.nchunks;
    const int clast = (nch - 1) * WKS;
    WinoPatch<MODE, 6, 4> pr;
    pr.template init<G::PH, G::PW, G::PWS, G::PLANE, true>(q.src, b, iy0, ix0, tid, clast, 2 * PF);
    // weights: [chunk][block64][row 4][position 6][half 2][plane 3][lane 64][8 bf16]; chunk / block / position / half in the scalar offset
    const auto wrs = wino_rsrc(p.w, (unsigned)((size_t)nch * q.nblk * W6S_BLK_BYTES));
    const unsigned wvo = (unsigned)(wave * 6 * W6S_POS_BYTES + lane * 16);
    const int wblk = nblk_i * W6S_BLK_BYTES, wchunk = q.nblk * W6S_BLK_BYTES;          // bytes
    u32x4 Aop[2][3];                                 // A operands (hi, mid, lo) of the position in flight and the one being built
    u32x4 Bop[4][2][3];                              // B operands [ring of four positions][32-channel half][plane]: requested three positions ahead
    float t[6][8];                                   // row `wave` of B2^T d: [column][channel] (columns of the next chunk replace dead ones)
    float4 qa[2][2], qb[2][2];                       // raw window rows of the (up to two) columns being read: [column slot][quad]
    float v[8], hf[8], sa[8], sd[8];                 // column-transform outputs of a position, unpacked bf16 terms, shared sub-expressions

    auto bload = [&](auto Pc, auto Rc, auto Kc, int chunk) {  // B operand k = plane * 2 + half (the order the MFMAs take them) of position P of `chunk` -> ring slot R
        constexpr int P = decltype(Pc)::value, R = decltype(Rc)::value, K = decltype(Kc)::value;
        // byte offset inside the position: half * 3072 + plane * 1024; two scalar bases per position (+ 0 / + 4096) and the rest in the
        // instruction's 12-bit immediate instead of one scalar addition per load
        constexpr int OFF = (K & 1) * 3072 + (K >> 1) * 1024, HI = OFF >= 4096 ? 4096 : 0;
        Bop[R][K & 1][K >> 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            wrs, (int)(wvo + (OFF - HI)), chunk * wchunk + wblk + P * W6S_POS_BYTES + HI, 0));
    };
    // ONE wait per position: the six B operands of ring slot r (requested three positions ago) pass through a common use in front of the
    // position's first MFMA, so that the compiler waits for all of them there instead of once in front of each of six MFMAs
    auto bready = [&](int r) {
#define W6S_BUSE(R_) asm volatile("" : "+v"(Bop[R_][0][0]), "+v"(Bop[R_][0][1]), "+v"(Bop[R_][0][2]), "+v"(Bop[R_][1][0]), "+v"(Bop[R_][1][1]), "+v"(Bop[R_][1][2]))
        if (r == 0) W6S_BUSE(0);
        else if (r == 1) W6S_BUSE(1);
        else if (r == 2) W6S_BUSE(2);
        else W6S_BUSE(3);
#undef W6S_BUSE
    };
    // t[j] of the NEXT chunk: 4 LDS reads (rows ra / rb x quads 2 hq, 2 hq + 1) and 8 combinations per column
    auto tread = [&](auto Jc, auto Sc, auto Kc, const float *pn) {
        constexpr int J = decltype(Jc)::value, S = decltype(Sc)::value, K = decltype(Kc)::value;
        if (K < 2) qa[S][K] = ld4(pn + pra + K * PL + J * 4);
        else qb[S][K - 2] = ld4(pn + prb + (K - 2) * PL + J * 4);
    };
    auto tcomb = [&](auto Jc, auto Sc, auto Ec) {
        constexpr int J = decltype(Jc)::value, S = decltype(Sc)::value, E = decltype(Ec)::value;
        const float4 x = qa[S][E >> 2], y = qb[S][E >> 2];
        const float xe = (E & 3) == 0 ? x.x : (E & 3) == 1 ? x.y : (E & 3) == 2 ? x.z : x.w;
        const float ye = (E & 3) == 0 ? y.x : (E & 3) == 1 ? y.y : (E & 3) == 2 ? y.z : y.w;
        t[J][E] = fmaf(sb, ye, xe);
        if (ABL6S(8)) ABL6S_KEEP(t[J][E]);
    };
    // instruction k of the column transform of position Q from row t
    //   B4^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
    auto colop = [&](auto Qc, auto Kc) {
        constexpr int Q = decltype(Qc)::value, K = decltype(Kc)::value, E = K & 7;
        struct Keep { float &x; __device__ ~Keep() { if (ABL6S(4) && K >= colops_n(Q) - 8) ABL6S_KEEP(x); } } keep{v[E]};
        if (Q == 0) {
            if (K < 8) v[E] = fmaf(-5.f, t[2][E], t[4][E]);
            else v[E] = fmaf(4.f, t[0][E], v[E]);
        } else if (Q == 1) {
            if (K < 8) sa[E] = fmaf(-4.f, t[2][E], t[4][E]);
            else if (K < 16) sd[E] = fmaf(-4.f, t[1][E], t[3][E]);
            else v[E] = sa[E] + sd[E];
        } else if (Q == 2) {
            v[E] = sa[E] - sd[E];
        } else if (Q == 3) {
            if (K < 8) sa[E] = t[4][E] - t[2][E];
            else if (K < 16) sd[E] = t[3][E] - t[1][E];
            else v[E] = fmaf(2.f, sd[E], sa[E]);
        } else if (Q == 4) {
            v[E] = fmaf(-2.f, sd[E], sa[E]);
        } else {
            if (K < 8) v[E] = fmaf(-5.f, t[3][E], t[5][E]);
            else v[E] = fmaf(4.f, t[1][E], v[E]);
        }
    };
    // instruction k (0..43) of the split of v[0..7] into ring slot R: per level 4 packed conversions, 8 unpacks, 8 exact residuals
    auto split = [&](auto Rc, auto Kc) {
        constexpr int R = decltype(Rc)::value, K = decltype(Kc)::value;
        constexpr int L = K / 20, KK = K % 20;       // level 0 (hi), 1 (mid), 2 (lo: the conversions only)
        if (KK < 4) {
            const unsigned pk = cvt_pk_bf16(v[2 * KK], v[2 * KK + 1]);
            if (KK == 0) Aop[R][L].x = pk;
            else if (KK == 1) Aop[R][L].y = pk;
            else if (KK == 2) Aop[R][L].z = pk;
            else Aop[R][L].w = pk;
        } else if (KK < 12) {
            constexpr int E = (KK - 4) & 7;
            const unsigned pk = (E >> 1) == 0 ? Aop[R][L].x : (E >> 1) == 1 ? Aop[R][L].y : (E >> 1) == 2 ? Aop[R][L].z : Aop[R][L].w;
            hf[E] = __uint_as_float((E & 1) ? (pk & 0xffff0000u) : (pk << 16));
        } else {
            constexpr int E = (KK - 12) & 7;
            v[E] = v[E] - hf[E];
        }
    };

    // ---- prologue: patch 0 -> LDS, B operands of the first three positions, row and first A operand of chunk 0.  (Requesting the patches of the
    // first TWO chunks together — one memory round trip instead of two — measured WORSE: first barrier 2.1 -> 2.6 us, the second 0.79 -> 0.72:
    // profiles/r06_probe_split_*.txt.)
    W6S_STAMP(1);
    pr.load(q.src, 0, clast);
    sfor<3>([&](auto P) { sfor<6>([&](auto K) { bload(P, P, K, 0); }); });
    pr.store(patch, q.src, 0);
    pr.load(q.src, min(WKS, clast), clast);
    __syncthreads();
    W6S_STAMP(2);
    sfor<6>([&](auto J) {
        sfor<4>([&](auto K) { tread(J, std::integral_constant<int, 0>{}, K, patch); });
        sfor<8>([&](auto E) { tcomb(J, std::integral_constant<int, 0>{}, E); });
    });
    sfor<16>([&](auto K) { colop(std::integral_constant<int, 0>{}, K); });
    sfor<44>([&](auto K) { split(std::integral_constant<int, 0>{}, K); });
    pr.store(patch + PF, q.src, min(WKS, clast));
    pr.load(q.src, min(2 * WKS, clast), clast);
    __syncthreads();
    W6S_STAMP(3);

    // ---- main loop.  One chunk = 6 positions x (6 products x 2 halves) MFMAs.  Behind every MFMA a slice of the position's OTHER work (seg_at):
    // the six B loads of the position THREE ahead; patch slot P of chunk + 2 (registers -> LDS) and its reload for chunk + 3; column transform
    // and split of the NEXT position's A operands (position 0 of the next chunk behind position 5); LDS reads and row combinations of the
    // next chunk's columns that have just died.  PAR = chunk & 1: six positions do not divide the ring of four.
    auto body = [&](auto par, int chunk) {
        constexpr int PAR = decltype(par)::value;
        const float *pnext = patch + ((chunk + 1) & 1) * PF;            // patch(chunk + 1)
        float *pfree = patch + (chunk & 1) * PF;                        // patch(chunk): consumed during chunk - 1 -> patch(chunk + 2)
        const int cw = min(chunk + 1, nch - 1);
        const int c2 = min((chunk + 2) * WKS, clast), c3 = min((chunk + 3) * WKS, clast);
        sfor<6>([&](auto Pc) {
            constexpr int P = decltype(Pc)::value, Q = (P + 1) % 6, N = ops_total(P);
            auto op = [&](auto Kc) {
                constexpr OpRef o = op_at(P, decltype(Kc)::value);
                using I = std::integral_constant<int, o.idx>;
                using J = std::integral_constant<int, o.col>;
                using S = std::integral_constant<int, o.slot>;
                if constexpr (o.kind == SEG_RD) {
                    if (!ABL6S(16)) tread(J{}, S{}, I{}, pnext);
                } else if constexpr (o.kind == SEG_COL) {
                    if (!ABL6S(8)) {
                        colop(std::integral_constant<int, Q>{}, I{});
                    }
                } else if constexpr (o.kind == SEG_SPLIT) {
                    if (!ABL6S(4)) split(std::integral_constant<int, Q & 1>{}, I{});
                } else {
                    if (!ABL6S(16)) tcomb(J{}, S{}, I{});
                }
            };
            if (!ABL6S(2)) bready((2 * PAR + P) & 3);
            sfor<12>([&](auto Mc) {
                constexpr int M = decltype(Mc)::value, PR = M >> 1, FH = M & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (!ABL6S(32))
                    // (weights as the MFMA's first operand: D = [channel][tile], a lane then holds FOUR CONSECUTIVE CHANNELS of its tile per
                    // register quad and the exchange below writes 16-byte cells)
                    acc[ABL6S(256) && (M & 2) ? (P + 3) % 6 : P][FH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, Bop[(2 * PAR + P) & 3][FH][prod_b(PR)]), __builtin_bit_cast(bf16x8, Aop[P & 1][prod_a(PR)]),
                        acc[ABL6S(256) && (M & 2) ? (P + 3) % 6 : P][FH], 0, 0, 0);       // (256: a tuning build's FOUR accumulators in flight)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((M & 1) == 0) {
                    if (!ABL6S(2)) bload(std::integral_constant<int, (P + 3) % 6>{}, std::integral_constant<int, (2 * PAR + P + 3) & 3>{},
                                         std::integral_constant<int, M / 2>{}, P + 3 < 6 ? chunk : cw);
                } else if constexpr (M == 1) {
                    if (!ABL6S(64)) pr.store_slot(pfree, q.src, c2, P);
                } else if constexpr (M == 3) {
                    if (!ABL6S(1)) pr.load_slot(q.src, c3, P, clast);
                }
                constexpr int LO = N * M / 12, HI = N * (M + 1) / 12;
                sfor<HI - LO>([&](auto Kc) { op(std::integral_constant<int, LO + decltype(Kc)::value>{}); });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                             // patch(chunk + 2) visible; patch(chunk + 1) free
    };
    int chunk = 0;
    do {
        body(std::integral_constant<int, 0>{}, chunk);
        if (chunk + 1 < nch) body(std::integral_constant<int, 1>{}, chunk + 1);      // (uniform over the workgroup)
        chunk += 2;
    } while (chunk < nch);
    W6S_STAMP(4);

    // ---- per 32-channel half: exchange (column transform M A4 of the wave's row, all waves -> LDS) and channel-quad epilogue as in conv_wino6.hip,
    // with one difference that matters for a workgroup ALONE on its CU: the epilogue's global operands (state, gates, old output: up to three
    // tensors x 8 pixels per thread) are requested BEFORE the exchange of their half, the second half's before the first half is finished — their
    // round trips (1-2.5 us each at four exposed requests per workgroup: profiles/r06_probe_split_before.txt) pass under the exchange, the barrier
    // and the other half's arithmetic instead of in front of every group of stores.
    // A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1];
    // D of the 32x32 MFMA (weights first): col = lane & 31 (TILE), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel of the 32-channel half):
    // registers 4 g .. 4 g + 3 are channels 8 g + 4 hq .. + 3 of the lane's tile — 16 ds_write_b128 per half instead of 64 ds_write_b32
    float *Pb = smem;
    const int qd = tid & 7;
    constexpr int NP = 8, RSTEP = 32 / RTW;                   // pixels per thread: one column, rows py0 + j * RSTEP
    const int px0 = (tid >> 3) % RTW, py0 = (tid >> 3) / RTW;
    const unsigned pix0 = (unsigned)((oy0 + py0) * p.WoF + ox0 + px0);      // inside image b (osy = osx = 1, no offsets: launcher)
    const size_t img = (size_t)b * p.HoF * p.WoF;
    const float *lbase = Pb + ((px0 & 3) * 32 + (px0 >> 2)) * RO_LD + qd * 4;
    auto rsrc_of = [&](const float *ptr, int ld) { return wino_rsrc(ptr ? ptr + img * ld : nullptr, WOOB); };
    auto bld = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
    auto bst = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off, float4 x) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), r, (int)off, 0, 0); };
    const int epi = p.epi;
    // 0: linear / ReLU (+ beta * old), 4: sigmoid, 5: sigmoid + h.r (gates), 1: residual + ReLU, 2: GRU blend, 3: GRU backward stage B, 6: out_s2d scatter
    const int kind = q.s2d_shift ? 6 : epi == RAMNET_EPI_GRU_BLEND ? 2 : epi == RAMNET_EPI_RES_RELU ? 1 : (epi == RAMNET_EPI_GRU_BWD && n0 >= p.Cout / 2) ? 3
                   : epi == RAMNET_EPI_SIGMOID ? 4 : epi == RAMNET_EPI_SIGMOID_HR ? 5 : 0;      // (GRU_BWD: a block lies in one half: launcher)
    const auto r_out = rsrc_of(p.out, p.ldo);
    const auto r_e0 = (kind >= 1 && kind <= 3) ? rsrc_of(p.e0, p.lde0) : r_out;
    const auto r_e1 = (kind == 2 || kind == 3 || kind == 5) ? rsrc_of(p.e1, p.lde1) : r_out;
    const auto r_o1 = (kind == 2 || kind == 3 || kind == 5) ? rsrc_of(p.o1, p.ldo1) : r_out;
    auto step_of = [&](int ld) { return (unsigned)(RSTEP * p.WoF * ld * 4); };
    const bool addold = kind == 0 && p.beta != 0.f && (epi == RAMNET_EPI_RELU || epi == RAMNET_EPI_LINEAR);
    const bool relu = epi == RAMNET_EPI_RELU;
    unsigned bad[NP];                                          // WOOB for the pixels of this thread outside the map (Cout % 64 == 0: every quad exists)
#pragma unroll
    for (int j = 0; j < NP; ++j) bad[j] = (ox0 + px0 < p.Wo && oy0 + py0 + j * RSTEP < p.Ho) ? 0u : WOOB;
    float4 ea[2][NP], eb[2][NP], ec[2][NP];                   // global operands of the two halves
    float4 bias4[2];
    auto nq_of = [&](int fh) { return n0 + fh * 32 + qd * 4; };
    auto off0_of = [&](int fh, int ld, bool have, int dn = 0) { return have ? (pix0 * (unsigned)ld + (unsigned)(nq_of(fh) + dn)) * 4u : WOOB; };
    // byte offsets at pixel 0 of the half's tensors: out, e0, e1, o1 (WOOB: no such operand for this thread)
    auto o_out = [&](int fh) { return off0_of(fh, p.ldo, true); };
    auto o_e0 = [&](int fh) { return off0_of(fh, p.lde0, kind >= 1 && kind <= 3); };
    auto o_e1 = [&](int fh) {
        return off0_of(fh, p.lde1, ((kind == 2 || kind == 3) && p.e1 != nullptr) || (kind == 5 && nq_of(fh) >= p.Cout / 2), (kind == 3 || kind == 5) ? -(p.Cout / 2) : 0);
    };
    auto o_o1 = [&](int fh) {
        return off0_of(fh, p.ldo1, ((kind == 2 || kind == 3) && p.o1 != nullptr) || (kind == 5 && nq_of(fh) >= p.Cout / 2), kind == 5 ? -(p.Cout / 2) : 0);
    };
    auto eload = [&](auto Fc) {                                // request the half's global operands (uniform branches on `kind`)
        constexpr int FH = decltype(Fc)::value;
        bias4[FH] = p.bias ? ld4(p.bias + nq_of(FH)) : f4zero();
        if (kind == 0 && addold) {
            const unsigned o = o_out(FH), st = step_of(p.ldo);
#pragma unroll
            for (int j = 0; j < NP; ++j) ea[FH][j] = bld(r_out, (o + j * st) | bad[j]);
        }
        if (kind >= 1 && kind <= 3) {
            const unsigned o = o_e0(FH), st = step_of(p.lde0);
#pragma unroll
            for (int j = 0; j < NP; ++j) ea[FH][j] = bld(r_e0, (o + j * st) | bad[j]);
        }
        if (kind == 2 || kind == 3 || kind == 5) {
            const unsigned o = o_e1(FH), st = step_of(p.lde1);
#pragma unroll
            for (int j = 0; j < NP; ++j) eb[FH][j] = bld(r_e1, (o + j * st) | bad[j]);
        }
        if (kind == 3) {
            const unsigned o = o_out(FH), st = step_of(p.ldo);
#pragma unroll
            for (int j = 0; j < NP; ++j) ec[FH][j] = bld(r_out, (o + j * st) | bad[j]);
        }
    };
    auto exchange = [&](auto Fc) {
        constexpr int FH = decltype(Fc)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float c0[4], c1[4], c2[4], c3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float m0 = acc[0][FH][r], m1 = acc[1][FH][r], m2 = acc[2][FH][r], m3 = acc[3][FH][r], m4 = acc[4][FH][r], m5 = acc[5][FH][r];
                const float a12 = m1 + m2, d12 = m1 - m2, a34 = m3 + m4, d34 = m3 - m4;
                c0[e] = m0 + a12 + a34, c1[e] = d12 + 2.f * d34, c2[e] = a12 + 4.f * a34, c3[e] = d12 + 8.f * d34 + m5;
            }
            float *dst = Pb + ((wave * 4) * 32 + l31) * RO_LD + FH * 32 + 8 * g + 4 * hq;
            st4(dst + 0 * 32 * RO_LD, make_float4(c0[0], c0[1], c0[2], c0[3]));
            st4(dst + 1 * 32 * RO_LD, make_float4(c1[0], c1[1], c1[2], c1[3]));
            st4(dst + 2 * 32 * RO_LD, make_float4(c2[0], c2[1], c2[2], c2[3]));
            st4(dst + 3 * 32 * RO_LD, make_float4(c3[0], c3[1], c3[2], c3[3]));
        }
    };
    // the row transform A2^T over the waves: even rows t0 + t1 + t2, odd rows t1 - t2 - t3 (conv_wino6.hip); then the activation of `kind`
    auto efinish = [&](auto Fc, auto Kc) {
        constexpr int FH = decltype(Fc)::value, K = decltype(Kc)::value;
        const int nq = nq_of(FH);
        const unsigned oo = o_out(FH), so = step_of(p.ldo), o1 = o_o1(FH), s1 = step_of(p.ldo1);
        const int g = K == 6 ? nq >> q.s2d_shift : 0;
        const unsigned fp0 = (unsigned)((2 * (oy0 + py0) + (g >> 1)) * p.WoF + 2 * (ox0 + px0) + (g & 1));      // (K = 6)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 t0[4], t1[4], t2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int py = py0 + (half * 4 + i) * RSTEP;
                const float *bb = lbase + ((py >> 1) * TXG + (py & 1) * 128) * RO_LD + FH * 32;
                t0[i] = ld4(bb), t1[i] = ld4(bb + 128 * RO_LD), t2[i] = ld4(bb + 256 * RO_LD);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = half * 4 + i;
                const float sg = ((py0 + j * RSTEP) & 1) ? -1.f : 1.f;
                float4 x = make_float4(fmaf(sg, t2[i].x, fmaf(sg, t1[i].x, t0[i].x)), fmaf(sg, t2[i].y, fmaf(sg, t1[i].y, t0[i].y)),
                                       fmaf(sg, t2[i].z, fmaf(sg, t1[i].z, t0[i].z)), fmaf(sg, t2[i].w, fmaf(sg, t1[i].w, t0[i].w)));
                if (K == 6) {     // out_s2d (backward-data of a stride-2 5x5 encoder over its space-to-depth view; LINEAR, no bias: checked on the host):
                    // channel quad nq of logical pixel (oy, ox) is channel quad nq - g * C of full-resolution pixel (2 oy + (g >> 1), 2 ox + (g & 1))
                    bst(r_out, ((fp0 * (unsigned)p.ldo + (unsigned)(nq - (g << q.s2d_shift))) * 4u + j * 2 * so) | bad[j], x);
                    continue;
                }
                x = f4add(x, bias4[FH]);
                if (K == 0) {
                    if (addold) x = make_float4(x.x + p.beta * ea[FH][j].x, x.y + p.beta * ea[FH][j].y, x.z + p.beta * ea[FH][j].z, x.w + p.beta * ea[FH][j].w);
                    if (relu) x = make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f));
                } else if (K == 4) {
                    x = make_float4(sigmoidf_(x.x), sigmoidf_(x.y), sigmoidf_(x.z), sigmoidf_(x.w));
                } else if (K == 5) {      // gates: the reset gate's quads also leave h.r (RAMNET_EPI_SIGMOID_HR; other quads: offset WOOB, h = 0)
                    x = make_float4(sigmoidf_(x.x), sigmoidf_(x.y), sigmoidf_(x.z), sigmoidf_(x.w));
                    const float4 h = eb[FH][j];
                    bst(r_o1, (o1 + j * s1) | bad[j], make_float4(h.x * x.x, h.y * x.y, h.z * x.z, h.w * x.w));
                } else if (K == 1) {
                    const float4 e = ea[FH][j];
                    x = make_float4(fmaxf(x.x + e.x, 0.f), fmaxf(x.y + e.y, 0.f), fmaxf(x.z + e.z, 0.f), fmaxf(x.w + e.w, 0.f));
                } else if (K == 3) {      // stage B of the ConvGRU backward on the d(h.r) half (RAMNET_EPI_GRU_BWD; conv_epilogue.hpp: gru_bwd_quad)
                    const float4 gq = x, r = ea[FH][j], h = eb[FH][j], old = ec[FH][j];
                    bst(r_o1, (o1 + j * s1) | bad[j], make_float4(gq.x * h.x * r.x * (1.0f - r.x), gq.y * h.y * r.y * (1.0f - r.y),
                                                                   gq.z * h.z * r.z * (1.0f - r.z), gq.w * h.w * r.w * (1.0f - r.w)));
                    x = make_float4(old.x + gq.x * r.x, old.y + gq.y * r.y, old.z + gq.z * r.z, old.w + gq.w * r.w);
                } else if (K == 2) {
                    const float4 o = make_float4(tanhf_(x.x), tanhf_(x.y), tanhf_(x.z), tanhf_(x.w)), u = ea[FH][j], h = eb[FH][j];
                    bst(r_o1, (o1 + j * s1) | bad[j], o);
                    x = make_float4(h.x * (1.0f - u.x) + o.x * u.x, h.y * (1.0f - u.y) + o.y * u.y, h.z * (1.0f - u.z) + o.z * u.z,
                                    h.w * (1.0f - u.w) + o.w * u.w);
                }
                bst(r_out, (oo + j * so) | bad[j], x);
            }
        }
    };
    auto finish = [&](auto Fc) {
        if (kind == 0) efinish(Fc, std::integral_constant<int, 0>{});
        else if (kind == 1) efinish(Fc, std::integral_constant<int, 1>{});
        else if (kind == 2) efinish(Fc, std::integral_constant<int, 2>{});
        else if (kind == 3) efinish(Fc, std::integral_constant<int, 3>{});
        else if (kind == 4) efinish(Fc, std::integral_constant<int, 4>{});
        else if (kind == 5) efinish(Fc, std::integral_constant<int, 5>{});
        else efinish(Fc, std::integral_constant<int, 6>{});
    };
    using F0 = std::integral_constant<int, 0>;
    using F1 = std::integral_constant<int, 1>;
    // ONE exchange for both 32-channel halves (136 KB of LDS: the workgroup is alone on its CU anyway): the operands of both halves are requested
    // first, one barrier instead of three, and the second half's LDS reads run under the first half's stores.
    eload(F0{});
    eload(F1{});
    exchange(F0{});
    exchange(F1{});
    __syncthreads();
    W6S_STAMP(5);
    finish(F0{});
    W6S_STAMP(6);
    W6S_STAMP(11);
    finish(F1{});
    W6S_STAMP(12);
#ifdef RAMNET_PROBE
    __builtin_amdgcn_s_waitcnt(0);          // (the stores have left the wave)
    W6S_STAMP(13);
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_probe6s[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

// OIHW 3x3 -> U = G2 g G4^T (evaluated in double, rounded to fp32 like the pack of conv_wino6.hip) split into three bf16 planes in the
// lane order of the kernel's B operand:
// 16-byte cell index = ((((((chunk * nblk + nb) * 4 + w) * 6 + pl) * 2 + f) * 3 + plane) * 64 + lane), its 8 bf16 = input channels
// chunk * 16 + 8 (lane >> 5) + 0..7 of U[row w][column pl][.][output channel nb * 64 + f * 32 + (lane & 31)]
__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void pack_weight_wino_r6s_kernel(const float *__restrict__ w, unsigned short *__restrict__ wp, int Cout, int Cin, int transposed,
                                            int R, int N, int nchunks, int nblk, size_t total) {
    // one thread per (chunk, nb, w, pl, f, lane, e): all three planes of one weight
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), f = (int)((i >> 9) & 1);
        size_t jj = i >> 10;
        const int pl = (int)(jj % 6);
        jj /= 6;
        const int wv = (int)(jj & 3);
        jj >>= 2;
        const int nb = (int)(jj % nblk), chunk = (int)(jj / nblk);
        const int r = chunk * WKS + 8 * (lane >> 5) + e;
        const int no = nb * W6S_BN + f * 32 + (lane & 31);
        float x = 0.f;
        if (r < R && no < N) {
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb)
                    g[a][bb] = transposed ? (double)w[((size_t)r * Cin + no) * 9 + (2 - a) * 3 + (2 - bb)]
                                          : (double)w[((size_t)no * Cin + r) * 9 + a * 3 + bb];
            const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const double G4[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                     {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) s += G2[wv][a] * g[a][bb] * G4[pl][bb];
            x = (float)s;
        }
        const size_t cell = ((((((size_t)chunk * nblk + nb) * 4 + wv) * 6 + pl) * 2 + f) * 3) * 64 + lane;     // plane 0
        const unsigned short h1 = bf16_rne(x);
        const float r1 = x - __uint_as_float((unsigned)h1 << 16);
        const unsigned short h2 = bf16_rne(r1);
        const float r2 = r1 - __uint_as_float((unsigned)h2 << 16);
        const unsigned short h3 = bf16_rne(r2);
        wp[(cell + 0 * 64) * 8 + e] = h1;
        wp[(cell + 1 * 64) * 8 + e] = h2;
        wp[(cell + 2 * 64) * 8 + e] = h3;
    }
}

static void wino6s_geometry(int Cout, int Cin, int transposed, int &R, int &N, int &nchunks, int &nblk) {
    R = transposed ? Cout : Cin;
    N = transposed ? Cin : Cout;
    nchunks = cdiv(R, WKS);
    nblk = cdiv(N, W6S_BN);
}

int wino6_eligible(const ramnet_conv_desc &d, int force);          // conv_wino6.hip

// The split-operand form runs what the exact-fp32 F(2x4,3x3) kernel runs, with 16-channel chunks: a chunk must lie in one tensor of a
// concatenation and in one parity group of a space-to-depth view.
int wino6s_eligible(const ramnet_conv_desc &d, int force) {
    if (!wino6_eligible(d, force)) return 0;
    if ((d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) && d.C0 % WKS != 0) return 0;
    if (d.in_mode == RAMNET_IN_S2D && d.C0 < WKS) return 0;
    return 1;
}

int launch_wino6s(const ramnet_conv_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 9 && d.stride == 1 && !d.frame && d.epi != RAMNET_EPI_LSTM && d.Cout % 64 == 0);
    RAMNET_CHECK_ARG(d.in_mode == RAMNET_IN_PLAIN || d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK ||
                     d.in_mode == RAMNET_IN_S2D);
    auto log2_exact = [](int x) { int sh = 0; while ((1 << sh) < x) ++sh; return (1 << sh) == x ? sh : -1; };
    auto al16 = [](const void *ptr) { return ptr == nullptr || ((uintptr_t)ptr & 15) == 0; };
    if (d.in_mode == RAMNET_IN_S2D) RAMNET_CHECK_ARG(d.C0 >= WKS && log2_exact(d.C0) > 0);                 // a chunk lies in one parity group
    if (d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) RAMNET_CHECK_ARG(d.C0 % WKS == 0);   // chunks do not straddle the concatenation
    RAMNET_CHECK_ARG(d.Cout % 4 == 0 && d.ldo % 4 == 0 && al16(d.out) && al16(d.bias) && (!d.o1 || (d.ldo1 % 4 == 0 && al16(d.o1))) &&
                     (!d.e0 || (d.lde0 % 4 == 0 && al16(d.e0))) && (!d.e1 || (d.lde1 % 4 == 0 && al16(d.e1))) && al16(d.w));
    RAMNET_CHECK_ARG(d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0);
    if (d.out_s2d) RAMNET_CHECK_ARG(d.out_s2d >= 8 && log2_exact(d.out_s2d) > 0 && d.Cout == 4 * d.out_s2d && d.epi == RAMNET_EPI_LINEAR && !d.bias &&
                                    d.beta == 0.f && d.HoF == 2 * d.Ho && d.WoF == 2 * d.Wo);
    if (d.epi == RAMNET_EPI_GRU_BWD) RAMNET_CHECK_ARG(d.Cout % 128 == 0);
    int dymin = 127, dxmin = 127;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin;
    }
    for (int t = 0; t < 9; ++t) {
        const int a = d.dy[t] - dymin, c = d.dx[t] - dxmin;
        RAMNET_CHECK_ARG(a >= 0 && a < 3 && c >= 0 && c < 3);
        seen |= 1u << (a * 3 + c);
    }
    RAMNET_CHECK_ARG(seen == 0x1ffu);
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    WinoParams q;
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    if (d.in_mode == RAMNET_IN_S2D) q.src.Cin = 4 * d.C0, q.src.ld1 = log2_exact(d.C0);
    q.nchunks = cdiv(q.src.Cin, WKS), q.nblk = cdiv(d.Cout, W6S_BN);
    // workgroup tile as in conv_wino6.hip: 16 x 16, 32 x 8 or 8 x 32 output pixels, whichever pads the map least
    int txg = 4;
    {
        const int shapes[3] = {4, 2, 8};
        long best = -1;
        for (int s = 0; s < 3; ++s) {
            const int t = shapes[s], th = 2 * (32 / t), tw = 4 * t;
            const long a = (long)cdiv(d.Ho, th) * th * cdiv(d.Wo, tw) * tw;
            if (best < 0 || a < best) best = a, txg = t;
        }
    }
    q.tiles_x = cdiv(d.Wo, 4 * txg), q.tiles_y = cdiv(d.Ho, 2 * (32 / txg));
    q.dy0 = dymin, q.dx0 = dxmin;
    q.vec4 = 1, q.s2d_shift = d.out_s2d ? log2_exact(d.out_s2d) : 0, q.sparse = 0;
    q.ksplit = 1, q.ws = nullptr, q.cnt = nullptr;
    // XCD-pinned channel groups for weights that do not fit an L2 (conv_wino.hip)
    const size_t wbytes = (size_t)q.nchunks * q.nblk * W6S_BLK_BYTES;
    q.xg = wbytes > (12u << 20) ? 2 : wbytes > (3u << 20) ? 1 : 0;
    while (q.xg > 0 && (q.nblk % (1 << q.xg)) != 0) --q.xg;
    const int lanes = 8 >> q.xg;
    q.inv_nbl = 1.0f / (float)(q.nblk >> q.xg), q.inv_tx = 1.0f / (float)q.tiles_x, q.inv_ty = 1.0f / (float)q.tiles_y;
    dim3 grid(cdiv(q.tiles_x * q.tiles_y * d.B, lanes) * 8 * (q.nblk >> q.xg));
    const size_t ex = (size_t)4 * 4 * 32 * (64 + 4) * sizeof(float);       // exchange buffer: both 32-channel halves
    {
        const unsigned long long px = (unsigned long long)d.Hin * d.Win * (d.in_mode == RAMNET_IN_S2D ? 4 : 1);
        int ldmax = d.ld0 > d.ld1 ? d.ld0 : d.ld1;
        ldmax = ldmax > d.ldm ? ldmax : d.ldm;
        RAMNET_CHECK_ARG(px * ldmax * 4ull < (unsigned long long)WOOB);        // per-image 32-bit byte offsets
        int lo = d.ldo > d.ldo1 ? d.ldo : d.ldo1;                              // ... of the epilogue's tensors too
        lo = lo > d.lde0 ? lo : d.lde0;
        lo = lo > d.lde1 ? lo : d.lde1;
        RAMNET_CHECK_ARG((unsigned long long)d.HoF * d.WoF * lo * 4ull < (unsigned long long)WOOB);
    }
    note_kernel("conv_wino_r6s_kernel<%d,%d>", txg, d.in_mode);
#define RAMNET_GO6S(TXv, MDv)                                                                                       \
    case (TXv) * 100 + (MDv): {                                                                                     \
        const size_t pf = (size_t)(2 * R6SGeom<TXv>::PFLOATS + 256 * 4) * sizeof(float);                            \
        RAMNET_FULL_LDS((conv_wino_r6s_kernel<TXv, MDv>));                                                          \
        hipLaunchKernelGGL((conv_wino_r6s_kernel<TXv, MDv>), grid, dim3(256), (ex > pf ? ex : pf), st, d, q);       \
    } break;
#if RAMNET_ABL6S
#define RAMNET_GO6S_TX(TXv) RAMNET_GO6S(TXv, RAMNET_IN_CAT)
#else
#define RAMNET_GO6S_TX(TXv)                                                                                         \
    RAMNET_GO6S(TXv, RAMNET_IN_PLAIN) RAMNET_GO6S(TXv, RAMNET_IN_CAT) RAMNET_GO6S(TXv, RAMNET_IN_CAT_MUL) RAMNET_GO6S(TXv, RAMNET_IN_RELUMASK) \
    RAMNET_GO6S(TXv, RAMNET_IN_S2D)
#endif
    switch (txg * 100 + d.in_mode) {
        RAMNET_GO6S_TX(4)
        RAMNET_GO6S_TX(2)
#if !RAMNET_ABL6S
        RAMNET_GO6S_TX(8)
#endif
    default:
        RAMNET_CHECK_ARG(!"conv_wino_r6s: unsupported (tile, input mode) combination");
    }
#undef RAMNET_GO6S_TX
#undef RAMNET_GO6S
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

#ifdef RAMNET_PROBE
extern "C" int ramnet_probe6s_read(unsigned long long *dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ramnet::g_probe6s), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int ramnet_conv_wino_split_ok(const ramnet_conv_desc *d, int force) { return d ? wino6s_eligible(*d, force) : 0; }

extern "C" size_t ramnet_packed_weight_elems_wino2x4_split(int Cout, int Cin, int transposed) {
    int R, N, nchunks, nblk;
    wino6s_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    return (size_t)nchunks * nblk * (W6S_BLK_BYTES / 4);           // in 4-byte units: the caller allocates floats
}

extern "C" int ramnet_pack_weight_wino2x4_split(const float *w, float *wp, int Cout, int Cin, int transposed, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0);
    int R, N, nchunks, nblk;
    wino6s_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    const size_t total = (size_t)nchunks * nblk * 4 * 6 * 2 * 64 * 8;          // weights (three planes each)
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_wino_r6s_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short *)wp, Cout, Cin,
                       transposed, R, N, nchunks, nblk, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
