// Backward-weights of the 3x3 stride-1 layers with SPLIT OPERANDS for gfx950 (MI355X): the launches of csrc/conv_wgrad_wino6.hip (ConvGRU
// gates / candidate, residual blocks, the space-to-depth views of the stride-2 encoders: submodules.py:447-452, 200-215) when the caller runs
// its 3x3 layers with three-term bf16 splits (ops.set_split_operands, csrc/conv_wino6s.hip).  DIRECT form, no Winograd transform:
//
//   dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co],     x = x1 + x2 + x3,  dy = g1 + g2 + g3   (bf16 terms, each the
//   round-to-nearest bf16 of what the terms before it left),  x dy ~= x1 g1 + x1 g2 + x2 g1 + x1 g3 + x2 g2 + x3 g1   (<= 2^-25 |x dy| dropped)
//
// 54 v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 pixels and 32 x 32 channel block = 1728 cycles of the matrix pipe, against 24
// v_mfma_f32_32x32x2_f32 (64 cycles each) = 1536 of the F(2x4,3x3) form — but that loop is ISSUE-bound (both Winograd transforms are built in
// registers by every wave: 13.8 other instructions per MFMA, 40-46 % MFMA-busy, profiles/r05_h_tuning_notes.md section 9), while here the operands
// are split ONCE per element when a strip is staged and the MFMA operands are plain LDS reads: 21 LDS reads + 12 v_alignbit per 54 MFMAs.
//
// Workgroup = 4 waves = 64 input x 64 output channels; wave (i, j) owns the 32 x 32 quadrant (input half i, output half j) of ALL nine taps
// (144 accumulators), two workgroups per CU.  A batch is 4 x 8 output pixels = two K steps of 16 pixels (lane group g = lane / 32 holds row
// 2 s + g, eight consecutive pixels): the A operand of tap (dy, dx) is the lane's window row shifted by dx pixels — one ds_read_b128 + one
// ds_read_b32 per (row, term) serve dx = 0 / 2 as register sub-ranges and dx = 1 through four v_alignbit_b32.  Strips are staged
// global -> registers (one batch ahead, under the MFMAs) -> split -> LDS [term][row][channel][pixels] as packed bf16 pairs.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.hpp"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Tuning builds only (tools/abl_dsplit.sh, -DRAMNET_ABLD=<mask>): one ingredient removed — 1 global loads, 2 split + LDS stores, 4 LDS operand reads
// and funnel shifts, 8 MFMAs, 16 barriers, 32 the join, 64 loads from one 32 KB window, 128 no funnel shifts / moves, 256 no LDS operand reads of the input strip.  Such a build computes WRONG results; only its duration means something.
#ifndef RAMNET_ABLD
#define RAMNET_ABLD 0
#endif
#define ABLD(bit) ((RAMNET_ABLD & (bit)) != 0)

namespace ramnet {

template <class F, int... I>
__device__ __forceinline__ void ds_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): every index a compile-time constant (the unroller gives up on the
// 108-MFMA iteration of the masked instantiations: accumulators indexed at run time end up in scratch)
template <int N, class F>
__device__ __forceinline__ void ds_for(F &&f) { ds_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ float g_dsplit_zero[4];                    // what an out-of-image / out-of-range slot loads (device globals start as zeros)

constexpr int DS_TARGET = 512;                        // workgroups per launch the tile splits aim at
// LDS, two buffers of [input strip: channel 64][row 6][term 3][6 dwords: 10 bf16 pixels + pad] + [gradient strip: term 3][row 4][channel 64][16 bytes].
// A lane reads a window row as two aligned ds_read_b64 + one ds_read_b32 (2 LDS cycles each, banks mod 64; ds_read2_b32 at an odd dword costs
// 4 cycles on 32 banks): dwords 0-3 are the dx = 0 operand, 1-4 the dx = 2 one (three register moves: MFMA operands are even-aligned register
// tuples), their 16-bit funnel shifts the dx = 1 one.  Channel pitch 110 dwords: 2 x an odd number, so that the 32 lanes of a row group cover
// all 64 banks with their 8-byte reads; the other row group is the other lane group of the instruction.  The gradient strip is read with
// aligned ds_read_b128.
constexpr int DS_TP = 24;                             // bytes of one (channel, row, term): 5 dwords + 1
constexpr int DS_RP = 3 * DS_TP;                      // bytes of one (channel, row)
constexpr int DS_CP = 6 * DS_RP + 8;                  // bytes of one channel of the input strip: 440
constexpr int DS_YCP = 16, DS_YRP = 64 * DS_YCP;      // gradient strip: channel / row pitch
constexpr int DS_YB = 64 * DS_CP;                     // gradient strip behind the input strip (28160)
constexpr int DS_YT = 4 * DS_YRP;                     // one term plane of the gradient strip
constexpr int DS_BUF = DS_YB + 3 * DS_YT;             // 40448 bytes per buffer
constexpr int DS_LDS = 2 * DS_BUF;                    // 80896 bytes: two workgroups per CU
struct __attribute__((packed, aligned(4))) DsU4 { u32x4 v; };      // four dwords at a dword-aligned LDS address (two ds_read2_b32)

struct WgradDsParams {
    InSrc src;
    int bx_n, ty_n, nbatch;     // strips per row, strip rows per image, total
    int dy0, dx0;               // offset of the first filter tap
    int splits, gy, gz, xcd_map;
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ds_cvt_pk(float lo, float hi) {      // {bf16(lo), bf16(hi)}, round to nearest even: v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{lo, hi}), bf16x2));
}
// a * b, rounded ONCE (HIP's __fmul_rn is a plain product: left to -ffp-contract=fast, the first residual of the split becomes fma(x, m, -x1))
__device__ __forceinline__ float ds_mul_rn(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// three bf16 terms of the pixel pair (a, b): h[t] = {term t of a, term t of b}
__device__ __forceinline__ void ds_split2(float a, float b, unsigned (&h)[3]) {
    h[0] = ds_cvt_pk(a, b);
    const float ra = a - __uint_as_float(h[0] << 16), rb = b - __uint_as_float(h[0] & 0xffff0000u);
    h[1] = ds_cvt_pk(ra, rb);
    h[2] = ds_cvt_pk(ra - __uint_as_float(h[1] << 16), rb - __uint_as_float(h[1] & 0xffff0000u));
}
// the six partial products (term of x, term of dy), grouped by the term of x so that one term's three shifted operands are live at a time
__device__ constexpr int ds_pa(int i) { return i == 0 ? 2 : i <= 2 ? 1 : 0; }
__device__ constexpr int ds_pb(int i) { return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 0 : i == 3 ? 2 : i == 4 ? 1 : 0; }

// XMK: second operand of the input loader — 0 none, 1 ReLU mask (x * (xm > 0)), 2 product (the second tensor of a CAT_MUL input); GM: ReLU mask on dy
template <int XMK, bool GM>
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wgrad_dsplit_kernel(const ramnet_wgrad_desc p, const WgradDsParams q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    const int ih = wave >> 1, jh = wave & 1;
    // workgroup -> (tile split, input block, output block): the gy x gz workgroups of one split read the same strips and sit on one XCD
    // (csrc/conv_wgrad_wino6.hip)
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
    const int c0 = by_i * 64, n0 = bz_i * 64;
    const InSrc &s = q.src;

    f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- staging slots: a slot = two neighbouring pixels x four channels.  Input strip 6 rows x 5 pairs x 16 quads = 480 slots (two per
    // thread, the second one for tid < 224), gradient strip 4 x 4 x 16 = 256 (one per thread); a thread's quad is the same in all of them.
    const int quad = tid & 15, pp = tid >> 4;
    const int cq = c0 + 4 * quad, nq = n0 + 4 * quad;
    const bool s2d = s.mode == RAMNET_IN_S2D;
    const bool cat = s.mode == RAMNET_IN_CAT || s.mode == RAMNET_IN_CAT_MUL;
    const bool second = cat && cq >= s.C0;
    const int pxs = s2d ? 2 : 1, WinS = pxs * s.Win, HinS = pxs * s.Hin;
    const float *xb, *mb = g_dsplit_zero;
    int ldx, ldmk = 0;
    if (s2d) {                                  // the quad's parity group (a, b): stored pixel (2 iy + a, 2 ix + b), s.ld1 = log2 C0
        const int g2 = cq >> s.ld1;
        xb = s.x0 + (cq - (g2 << s.ld1)) + (long)((g2 >> 1) * WinS + (g2 & 1)) * s.ld0, ldx = s.ld0;
    } else if (second) {
        xb = s.x1 + (cq - s.C0), ldx = s.ld1;
    } else {
        xb = s.x0 + cq, ldx = s.ld0;
    }
    bool use_m = false;
    if (XMK == 1) mb = s.xm + cq, ldmk = s.ldm, use_m = true;
    if (XMK == 2 && second) mb = s.xm + (cq - s.C0), ldmk = s.ldm, use_m = true;
    const float m_one = (XMK == 2 && !use_m) ? 1.f : 0.f;
    const bool cok = cq < s.Cin, nok = nq < p.Cout;
    int xrow[2], xpair[2], xpo[2], xdst[2];
    bool xvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ppi = pp + 16 * i;
        xvalid[i] = true;
        const int pps = ppi < 30 ? ppi : pp;     // (threads without a second slot repeat their first one: the same values to the same cells, no branch)
        xrow[i] = pps / 5, xpair[i] = pps - 5 * xrow[i];
        xpo[i] = pxs * xrow[i] * WinS + pxs * 2 * xpair[i];
        xdst[i] = 4 * quad * DS_CP + xrow[i] * DS_RP + 4 * xpair[i];
    }
    const int yrow = pp >> 2, ypair = pp & 3;
    const int ypo = yrow * p.Wo + 2 * ypair;
    const int ydst = DS_YB + yrow * DS_YRP + 4 * quad * DS_YCP + 4 * ypair;

    float4 xv[2][2], xmv[2][2], yv[2], ymv[2];
    float4 bsum = f4zero();
    float bias_on = 1.f;                        // 0 for the clamped re-staging of the last batch
    int lb_ty = 0, lb_bx = 0, lb_b = 0;
    int oy0 = 0, ox0 = 0;
    long pix0 = 0;
    auto coords = [&]() {
        oy0 = 4 * lb_ty, ox0 = 8 * lb_bx;
        pix0 = ((long)lb_b * HinS + pxs * (oy0 + q.dy0)) * WinS + pxs * (ox0 + q.dx0);
    };
    auto load_x = [&](int i) {                  // global -> registers: slot i of the input strip of batch (lb_b, lb_ty, lb_bx)
        if (ABLD(1)) {
#pragma unroll
            for (int k = 0; k < 2; ++k) asm volatile("" : "=v"(xv[i][k].x), "=v"(xv[i][k].y), "=v"(xv[i][k].z), "=v"(xv[i][k].w));
            return;
        }
        const int iy = oy0 + q.dy0 + xrow[i], ix = ox0 + q.dx0 + 2 * xpair[i];
        const bool rok = xvalid[i] & cok & ((unsigned)iy < (unsigned)s.Hin);
        const bool ok0 = rok & ((unsigned)ix < (unsigned)s.Win), ok1 = rok & ((unsigned)(ix + 1) < (unsigned)s.Win);
        const long pix = ABLD(64) ? ((pix0 + xpo[i]) & 63) : pix0 + xpo[i];
        const float *a = xb + pix * ldx;
        xv[i][0] = ld4(ok0 ? a : g_dsplit_zero);
        xv[i][1] = ld4(ok1 ? a + pxs * ldx : g_dsplit_zero);
        if (XMK) {
            const long lp = ((long)lb_b * s.Hin + iy) * s.Win + ix;          // masks are plain [B][Hin][Win][.] tensors
            const float *m = mb + lp * ldmk;
            xmv[i & 1][0] = ld4((ok0 & use_m) ? m : g_dsplit_zero);
            xmv[i & 1][1] = ld4((ok1 & use_m) ? m + ldmk : g_dsplit_zero);
        }
    };
    auto load_y = [&]() {
        if (ABLD(1)) {
#pragma unroll
            for (int k = 0; k < 2; ++k) asm volatile("" : "=v"(yv[k].x), "=v"(yv[k].y), "=v"(yv[k].z), "=v"(yv[k].w));
            return;
        }
        const int oy = oy0 + yrow, ox = ox0 + 2 * ypair;
        const bool rok = nok & (oy < p.Ho);
        const bool ok0 = rok & (ox < p.Wo), ok1 = rok & (ox + 1 < p.Wo);
        const long pix = ABLD(64) ? ((((long)lb_b * p.Ho + oy0) * p.Wo + ox0 + ypo) & 63) : ((long)lb_b * p.Ho + oy0) * p.Wo + ox0 + ypo;
        const float *a = p.dout + nq + pix * p.ldg;
        yv[0] = ld4(ok0 ? a : g_dsplit_zero);
        yv[1] = ld4(ok1 ? a + p.ldg : g_dsplit_zero);
        if (GM) {
            const float *m = p.gmask + nq + pix * p.ldgm;
            ymv[0] = ld4(ok0 ? m : g_dsplit_zero);
            ymv[1] = ld4(ok1 ? m + p.ldgm : g_dsplit_zero);
        }
    };
    auto advance = [&]() {                      // the walk over (image, strip row, strip) is an increment with carries
        ++lb_bx;
        if (lb_bx >= q.bx_n) {
            lb_bx = 0;
            if (++lb_ty >= q.ty_n) lb_ty = 0, ++lb_b;
        }
    };
    auto comp = [](const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
    // registers -> split -> LDS buffer `buf`: channel j of input slot i / of the gradient slot (even and odd pixel of the pair: one dword per term)
    auto stage_x = [&](int i, int j, unsigned char *buf) {
        if (ABLD(2)) return;
        float e = comp(xv[i][0], j), o = comp(xv[i][1], j);
        if (XMK == 1) e = comp(xmv[i & 1][0], j) > 0.f ? e : 0.f, o = comp(xmv[i & 1][1], j) > 0.f ? o : 0.f;
        if (XMK == 2) e = ds_mul_rn(e, comp(xmv[i & 1][0], j) + m_one), o = ds_mul_rn(o, comp(xmv[i & 1][1], j) + m_one);
        unsigned h[3];
        ds_split2(e, o, h);
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<unsigned *>(buf + xdst[i] + j * DS_CP + t * DS_TP) = h[t];
    };
    auto stage_y = [&](int j, unsigned char *buf) {
        if (ABLD(2)) return;
        float e = comp(yv[0], j), o = comp(yv[1], j);
        if (GM) e = comp(ymv[0], j) > 0.f ? e : 0.f, o = comp(ymv[1], j) > 0.f ? o : 0.f;
        unsigned h[3];
        ds_split2(e, o, h);
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<unsigned *>(buf + ydst + j * DS_YCP + t * DS_YT) = h[t];
        const float sm = bias_on * (e + o);
        if (j == 0) bsum.x += sm;
        if (j == 1) bsum.y += sm;
        if (j == 2) bsum.z += sm;
        if (j == 3) bsum.w += sm;
    };

    // ---- operands of the lane: channel 32 ih + l31 (input) / 32 jh + l31 (output), row group g; buffer, row and term are constant offsets
    const unsigned xop = (32 * ih + l31) * DS_CP + g * DS_RP;
    const unsigned char *yop = smem + DS_YB + g * DS_YRP + (32 * jh + l31) * DS_YCP;

    int batch = (int)((long long)q.nbatch * sp_i / q.splits);
    const int last = (int)((long long)q.nbatch * (sp_i + 1) / q.splits) - 1;
    if (batch <= last) {
        {
            int tt = batch;
            lb_bx = tt % q.bx_n;
            tt /= q.bx_n;
            lb_ty = tt % q.ty_n;
            lb_b = tt / q.ty_n;
        }
        coords();
        load_x(0), load_x(1), load_y();
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_x(0, j, smem), stage_x(1, j, smem), stage_y(j, smem);
        if (batch < last) advance();
        coords();
        load_x(0), load_x(1), load_y();
        __syncthreads();
        // One batch; CUR = the LDS buffer holding it, a compile-time constant (the loop alternates two instantiations).  The registers hold batch
        // + 1 when the iteration starts: its twelve (slot, channel) split-and-store steps and the three loads of batch + 2 — each right behind
        // the last step that read its registers, so that every load has a whole iteration to land — sit one per (K step, row, term) group
        // of MFMAs.
        auto iter = [&](auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            unsigned xo = xop + (ABLD(512) ? 0 : cur) * DS_BUF;
            asm volatile("" : "+v"(xo));          // (ONE base register per buffer: every read of the iteration is base + a small constant)
            const unsigned char *xc = smem + xo, *yc = yop + (ABLD(512) ? 0 : cur) * DS_BUF;
            unsigned char *nb = smem + (ABLD(512) ? 0 : (cur ^ 1)) * DS_BUF;
            bias_on = batch + 1 <= last ? 1.f : 0.f;
            u32x4 bop[3], a1, a2;
            uint2 lo[3], mid[3];                  // the raw window row of group gi in set gi % 3 (dwords 0-1, 2-3, 4), requested two groups ahead
            unsigned hi[3];
            auto fetch_b = [&](int ks) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (ABLD(4)) asm volatile("" : "=v"(bop[t]));
                    else bop[t] = *reinterpret_cast<const u32x4 *>(yc + t * DS_YT + 2 * ks * DS_YRP);
                }
            };
            auto fetch_a = [&](int gi) {              // group gi = (K step, row, term)
                const int ks = gi / 9, dyi = (gi / 3) % 3, ta = 2 - gi % 3, o = gi % 3;
                const int off = (2 * ks + dyi) * DS_RP + ta * DS_TP;
                if (ABLD(4) || ABLD(256)) asm volatile("" : "=v"(lo[o].x), "=v"(lo[o].y), "=v"(mid[o].x), "=v"(mid[o].y), "=v"(hi[o]));
                else lo[o] = *reinterpret_cast<const uint2 *>(xc + off), mid[o] = *reinterpret_cast<const uint2 *>(xc + off + 8),
                     hi[o] = *reinterpret_cast<const unsigned *>(xc + off + 16);
            };
            // ---- the staging work of an iteration as 57 slices of 3-5 instructions: the twelve (slot, channel) split-and-store steps of batch + 1
            // (four slices each: first term, second term, third term, the three stores) and the three loads of batch + 2, each right behind the
            // last step that read its registers (two slices + one spare)
            float se = 0.f, so = 0.f;
            unsigned sh[3] = {0u, 0u, 0u};
            auto stage_slice = [&](int kind, int i, int j, int part) {        // kind 0: input slot i, 1: gradient slot; channel j
                if (ABLD(2)) return;
                if (part == 0) {
                    if (kind == 0) {
                        se = comp(xv[i & 1][0], j), so = comp(xv[i & 1][1], j);
                        if (XMK == 1) se = comp(xmv[i & 1][0], j) > 0.f ? se : 0.f, so = comp(xmv[i & 1][1], j) > 0.f ? so : 0.f;
                        // (the SINGLE-rounded product the materialised h.r holds)
                        if (XMK == 2) se = ds_mul_rn(se, comp(xmv[i & 1][0], j) + m_one), so = ds_mul_rn(so, comp(xmv[i & 1][1], j) + m_one);
                    } else {
                        se = comp(yv[0], j), so = comp(yv[1], j);
                        if (GM) se = comp(ymv[0], j) > 0.f ? se : 0.f, so = comp(ymv[1], j) > 0.f ? so : 0.f;
                        const float sm = bias_on * (se + so);
                        if (j == 0) bsum.x += sm;
                        if (j == 1) bsum.y += sm;
                        if (j == 2) bsum.z += sm;
                        if (j == 3) bsum.w += sm;
                    }
                    sh[0] = ds_cvt_pk(se, so);
                } else if (part == 1 || part == 2) {
                    se -= __uint_as_float(sh[part - 1] << 16), so -= __uint_as_float(sh[part - 1] & 0xffff0000u);
                    sh[part] = ds_cvt_pk(se, so);
                } else {
                    unsigned char *d = nb + (kind == 0 ? xdst[i] + j * DS_CP : ydst + j * DS_YCP);
#pragma unroll
                    for (int t = 0; t < 3; ++t) *reinterpret_cast<unsigned *>(d + t * (kind == 0 ? DS_TP : DS_YT)) = sh[t];
                }
            };
            auto slice = [&](int idx) {
                if (idx < 0 || idx >= 57) return;
                const int blk = idx / 19, r = idx % 19;       // blocks: input slot 0, input slot 1, gradient slot
                if (r < 16) stage_slice(blk == 2 ? 1 : 0, blk, r / 4, r % 4);
                else if (r == 16) {
                    if (blk == 0) {
                        if (batch + 2 <= last) advance();       // (uniform)
                        coords();
                    }
                } else if (r == 17) {
                    if (blk == 0) load_x(0);
                    if (blk == 1) load_x(1);
                    if (blk == 2) load_y();
                }
            };
            fetch_b(0);
            fetch_a(0);
            fetch_a(1);
            ds_for<18>([&](auto gi_c) {
                constexpr int gi = decltype(gi_c)::value;
                constexpr int dyi = (gi / 3) % 3, ta = 2 - gi % 3, o = gi % 3;
                constexpr int m = ta == 2 ? 3 : ta == 1 ? 6 : 9;
                constexpr int free0 = 12 * (gi / 3) + (gi % 3 == 0 ? 0 : gi % 3 == 1 ? 1 : 5);      // free slots of the groups before this one
                const u32x4 a0 = u32x4{lo[o].x, lo[o].y, mid[o].x, mid[o].y};
                ds_for<m>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    // the k-th MFMA of the group: product k / 3 of this term, shift k % 3
                    int pr = -1, cnt = 0;
#pragma unroll
                    for (int q6 = 0; q6 < 6; ++q6)
                        if (ds_pa(q6) == ta) {
                            if (cnt == k / 3) pr = q6;
                            ++cnt;
                        }
                    const int dx = k % 3;
                    if (ABLD(8)) {
                        asm volatile("" : : "v"(a0), "v"(a1), "v"(a2), "v"(bop[ds_pb(pr)]));
                    } else {
                        const bf16x8 bb = __builtin_bit_cast(bf16x8, bop[ds_pb(pr)]);
                        const bf16x8 aa = __builtin_bit_cast(bf16x8, dx == 0 ? a0 : dx == 1 ? a1 : a2);
                        acc[dyi * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, bb, acc[dyi * 3 + dx], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- what goes behind it
                    if (k == 0) {             // the raw row of the group after the next one, this group's dx = 1 operand
                        if (gi + 2 < 18) fetch_a(gi + 2);
                        if (ABLD(4)) asm volatile("" : "=v"(a1));
                        else if (ABLD(128)) a1 = a0;
                        else a1 = u32x4{__builtin_amdgcn_alignbit(lo[o].y, lo[o].x, 16),
                                        __builtin_amdgcn_alignbit(mid[o].x, lo[o].y, 16), __builtin_amdgcn_alignbit(mid[o].y, mid[o].x, 16),
                                        __builtin_amdgcn_alignbit(hi[o], mid[o].y, 16)};
                    } else if (k == 1) {      // ... and its dx = 2 operand (an even-aligned register tuple: three moves)
                        if (ABLD(4)) asm volatile("" : "=v"(a2));
                        else if (ABLD(128)) a2 = a0;
                        else a2 = u32x4{lo[o].y, mid[o].x, mid[o].y, hi[o]};
                    } else {
                        slice(free0 + k - 2);
                        if (gi == 8 && k == m - 1) fetch_b(1);          // (every MFMA of K step 0 has been issued)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if (!ABLD(16)) __syncthreads();
        };
        do {                                        // (batch <= last here; ONE exit, at the bottom: csrc/conv_wgrad_wino6.hip)
            iter(std::integral_constant<int, 0>{});
            ++batch;
            if (batch <= last) {                    // (uniform)
                iter(std::integral_constant<int, 1>{});
                ++batch;
            }
        } while (batch <= last);
    }

    // ---- join: D of tap t -> the BLOCKED workspace [9 taps][Cin / 32][Cout / 32][64 lanes][16 accumulator registers] (the layout of
    // csrc/conv_wgrad_wino6.hip with 9 positions: a lane's 16 values — rows c = (r & 3) + 8 (r >> 2) + 4 g of column n = lane & 31 — are 64
    // contiguous bytes; ramnet_unpack_wgrad_dsplit reads it).  dw_slabs > 0: the split's own slab, plain read-modify-write (bit-reproducible),
    // tap t + 1 requested before t is stored; else atomics.
    const int nCiB = (s.Cin + 31) >> 5, nCoB = (p.Cout + 31) >> 5;
    const int cb = 2 * by_i + ih, nb = 2 * bz_i + jh;
    const bool slabs = p.dw_slabs > 0;
    if (cb < nCiB && nb < nCoB && (!ABLD(32) || acc[0][0] == 123.456f)) {
        const size_t pos_stride = (size_t)nCiB * nCoB * 1024;
        float *dwb = p.dw + (slabs ? (size_t)sp_i * 9 * pos_stride : 0) + ((size_t)cb * nCoB + nb) * 1024 + lane * 16;
        if (slabs) {
            float4 old[2][4];
            auto grp_load = [&](int t, float4 (&o)[4]) {
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] = ld4(dwb + (size_t)t * pos_stride + 4 * v);
            };
            grp_load(0, old[0]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) grp_load(t + 1, old[(t + 1) & 1]);
                float *d = dwb + (size_t)t * pos_stride;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 o = old[t & 1][v];
                    st4(d + 4 * v, make_float4(acc[t][4 * v] + o.x, acc[t][4 * v + 1] + o.y, acc[t][4 * v + 2] + o.z, acc[t][4 * v + 3] + o.w));
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(dwb + (size_t)t * pos_stride + r, acc[t][r]);
        }
    }
    if (p.dbias != nullptr && by_i == 0) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);          // [16][64]
        st4(red + pp * 64 + quad * 4, bsum);
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int k = 0; k < 16; ++k) t += red[k * 64 + tid];
            if (n0 + tid < p.Cout) {
                if (slabs) p.dbias[(size_t)sp_i * p.Cout + n0 + tid] += t;
                else atomicAdd(p.dbias + n0 + tid, t);
            }
        }
    }
}

bool wgrad_dsplit_eligible(const ramnet_wgrad_desc &d) {
    return d.ntaps == 9 && d.stride == 1 && d.Ho == d.Hin && d.Wo == d.Win && d.nseg == 0 &&
           (d.in_mode == RAMNET_IN_PLAIN || d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK ||
            (d.in_mode == RAMNET_IN_S2D && d.C0 >= 4 && (d.C0 & (d.C0 - 1)) == 0));
}

static int dsplit_splits(int Cin, int Cout) {
    static const int target = getenv("RAMNET_DS_TARGET") ? atoi(getenv("RAMNET_DS_TARGET")) : DS_TARGET;      // (tuning runs)
    const int s = target / (cdiv(Cin, 64) * cdiv(Cout, 64));
    return s < 1 ? 1 : s;
}

int launch_wgrad_dsplit(const ramnet_wgrad_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(wgrad_dsplit_eligible(d));
    int dymin = 127, dxmin = 127;
    for (int t = 0; t < 9; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin;
    }
    for (int t = 0; t < 9; ++t) RAMNET_CHECK_ARG(d.dy[t] - dymin == t / 3 && d.dx[t] - dxmin == t % 3);      // the forward tap order kh*3 + kw
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    WgradDsParams q;
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
    q.bx_n = cdiv(d.Wo, 8), q.ty_n = cdiv(d.Ho, 4), q.nbatch = q.bx_n * q.ty_n * d.B;
    const int gy = cdiv(q.src.Cin, 64), gz = cdiv(d.Cout, 64);
    int splits = dsplit_splits(q.src.Cin, d.Cout);
    if (splits > q.nbatch) splits = q.nbatch;
    if (d.dw_slabs > 0 && splits > d.dw_slabs) splits = d.dw_slabs;
    if (splits < 1) splits = 1;
    q.xcd_map = 0;
    if (splits >= 8) splits = splits / 8 * 8, q.xcd_map = 1;
    q.splits = splits, q.gy = gy, q.gz = gz;
    const dim3 grid(splits * gy * gz);
    const int xmk = d.in_mode == RAMNET_IN_RELUMASK ? 1 : d.in_mode == RAMNET_IN_CAT_MUL ? 2 : 0;
    const bool gm = d.gmask != nullptr;
    note_kernel("conv_wgrad_dsplit_kernel<%d,%d>", xmk, (int)gm);
#define RAMNET_GO(XMv, GMv)                                                                                          \
    do {                                                                                                             \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            RAMNET_FULL_LDS((conv_wgrad_dsplit_kernel<XMv, GMv>));                                                    \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((conv_wgrad_dsplit_kernel<XMv, GMv>), grid, dim3(256), (size_t)(ABLD(512) ? DS_BUF : DS_LDS), st, d, q);          \
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

// blocked ws [9][CinWs / 32][CoutWs / 32][64][16] (layout: the kernel's join) -> grad OIHW [Cout][Cin][3][3] (+=)
__global__ void unpack_wgrad_dsplit_kernel(const float *__restrict__ ws, float *__restrict__ g, int Cout, int Cin, int CinWs, int CoutWs, int n_off,
                                           size_t total) {
    const int nCiB = (CinWs + 31) / 32, nCoB = (CoutWs + 31) / 32;
    const size_t pos_stride = (size_t)nCiB * nCoB * 1024;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin), n = (int)(i / Cin);
        const int nn = n_off + n, c32 = c & 31;
        const size_t at = ((size_t)(c >> 5) * nCoB + (nn >> 5)) * 1024 + ((nn & 31) + 32 * ((c32 >> 2) & 1)) * 16 + (c32 & 3) + 4 * (c32 >> 3);
        for (int t = 0; t < 9; ++t) g[((size_t)n * Cin + c) * 9 + t] += ws[(size_t)t * pos_stride + at];
    }
}

}  // namespace ramnet
using namespace ramnet;

extern "C" int ramnet_wgrad_dsplit_slabs(int Cin, int Cout) { return dsplit_splits(Cin, Cout); }

extern "C" size_t ramnet_wgrad_dsplit_ws_floats(int Cin, int Cout) { return (size_t)9 * cdiv(Cin, 32) * cdiv(Cout, 32) * 1024; }

extern "C" int ramnet_unpack_wgrad_dsplit(const float *ws, float *grad, int Cout, int Cin, int CinWs, int CoutWs, int n_off, void *stream) {
    RAMNET_CHECK_ARG(ws && grad && Cout > 0 && Cin > 0 && CinWs >= Cin && n_off >= 0 && CoutWs >= n_off + Cout);
    const size_t total = (size_t)Cout * Cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(unpack_wgrad_dsplit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, grad, Cout, Cin, CinWs, CoutWs, n_off, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
