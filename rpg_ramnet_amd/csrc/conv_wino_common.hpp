// Shared by the Winograd kernels of the 3x3 layers (conv_wino.hip: F(2x2,3x3); conv_wino6.hip: F(2x4,3x3)): launch parameters and the
// branch-free patch prefetcher.
#pragma once
#include "common.hpp"

namespace ramnet {

constexpr int WK = 8;                         // input channels per chunk

struct WinoParams {
    InSrc src;
    int nchunks, nblk;      // Cin/8, CoutPad/64
    int tiles_x, tiles_y;
    int dy0, dx0;           // offset of the first filter tap (-1 for the padded 3x3)
    int vec4;               // all epilogue operands allow 16-byte channel-quad accesses
    int s2d_shift;          // log2(out_s2d) or 0
    int xg;                 // log2 of the number of XCD-pinned channel-block groups
    int sparse;             // ramnet_conv_desc.s2d_5x5: 1 = zero slices by input parity group (forward), 2 = by output group (backward-data)
    int ksplit;             // F(2x2) kernel, 32-channel workgroups: gridDim.y = splits of the channel reduction (1 = none)
    float *ws;              // ksplit > 1: partial output tiles [gridDim.x][ksplit][128 pixels][32 channels]
    int *cnt;               //             arrival counters [gridDim.x], zero between launches
    float inv_nbl, inv_tx, inv_ty;      // conv_wino6s.hip: reciprocals of (nblk >> xg), tiles_x, tiles_y (workgroup index decomposition without integer division)
};

// Patch prefetcher of the Winograd kernel, MODE = ramnet_in_mode of the launch as a compile-time constant (run-time, wave-uniform
// branches between the MFMAs are not free: measured 9 % for 32 scalar branch pairs per chunk).  Everything that does not depend
// on the chunk is computed once: the byte offsets of the thread's two (pixel, channel quad) slots inside image b of each source
// tensor — WOOB when the pixel lies outside the image, so that the buffer load returns the zero padding — and buffer resources
// over that image.  Per chunk a slot is ONE buffer load (+ one for the mask / h*r operand) with a scalar byte offset for the
// channel (and, for the space-to-depth view, the parity pixel), and one 16-byte LDS store.
// NS = (pixel, channel quad) slots per thread: 2 for the 128-pixel tiles of F(2x2,3x3), 3 for the 256-pixel tiles of F(2x4,3x3).
// QPP = channel quads per pixel and chunk: 2 (8-channel chunks, fp32 MFMA K steps) or 4 (16-channel chunks of the split-operand kernel,
// conv_wino6s.hip: one v_mfma_f32_32x32x16_bf16 reduces 16 channels).
template <int MODE, int NS = 2, int QPP = 2>
struct WinoPatch {
    static_assert(QPP == 2 || QPP == 4, "quads per pixel");
    static constexpr bool CAT = MODE == RAMNET_IN_CAT || MODE == RAMNET_IN_CAT_MUL;
    float4 v[NS], m[NS];
    unsigned vo0[NS], vo1[NS], vom[NS];      // byte offsets of (pixel, quad) in image b of x0 / x1 / xm, or WOOB
    unsigned bad[NS];                     // WOOB if the slot's quad lies beyond Cin in the LAST chunk (ragged channel counts), else 0
    int ldst[NS];                         // LDS float offset of the slot (a scratch location for the threads without one)
    decltype(wino_rsrc(nullptr, 0u)) r0, r1, rm;

    // LDS patch layout: [channel quad QPP][PH x PW pixels][4]; scratch = float offset of 256 spare 16-byte cells
    // PWS / PLANE / SKEW: row pitch in pixels, floats per quad plane, and a per-row column skew of ((row >> 1) & 3) pixels — the
    // bank-conflict-free layout of conv_wino6.hip (defaults: the dense layout of conv_wino.hip)
    template <int PH, int PW, int PWS = PW, int PLANE = PH * PW * 4, bool SKEW = false>
    __device__ __forceinline__ void init(const InSrc &s, int b, int iy0, int ix0, int tid, int clast, int scratch) {
        const bool s2d = MODE == RAMNET_IN_S2D;
        const size_t img = (size_t)(s2d ? 4 : 1) * s.Hin * s.Win;      // pixels of one image as stored
        r0 = wino_rsrc(s.x0 + (size_t)b * img * s.ld0, (unsigned)(img * s.ld0 * 4));
        r1 = r0, rm = r0;
        if (CAT) r1 = wino_rsrc(s.x1 + (size_t)b * img * s.ld1, (unsigned)(img * s.ld1 * 4));
        if (MODE == RAMNET_IN_CAT_MUL || MODE == RAMNET_IN_RELUMASK) rm = wino_rsrc(s.xm + (size_t)b * img * s.ldm, (unsigned)(img * s.ldm * 4));
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int sl = tid + i * 256;
            const int pix = sl / QPP, qd = sl % QPP;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool slot = sl < PH * PW * QPP;
            const bool in = slot && (unsigned)iy < (unsigned)s.Hin && (unsigned)ix < (unsigned)s.Win;
            // space-to-depth view: logical pixel (iy, ix) starts at full-resolution pixel (2iy, 2ix); the parity group of a
            // chunk only moves the (wave-uniform) scalar offset, see load_slot
            const unsigned gp = s2d ? (unsigned)((2 * iy) * 2 * s.Win + 2 * ix) : (unsigned)(iy * s.Win + ix);
            vo0[i] = in ? (gp * s.ld0 + qd * 4) * 4u : WOOB;
            vo1[i] = in ? (gp * s.ld1 + qd * 4) * 4u : WOOB;
            vom[i] = in ? (gp * s.ldm + qd * 4) * 4u : WOOB;
            bad[i] = clast + qd * 4 < s.Cin ? 0u : WOOB;
            ldst[i] = slot ? qd * PLANE + (py * PWS + px + (SKEW ? ((py >> 1) & 3) : 0)) * 4 : scratch + tid * 4;
        }
    }
    static __device__ __forceinline__ float4 bload(decltype(wino_rsrc(nullptr, 0u)) r, unsigned vo, int so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, so, 0));
    }
    // issue the loads of slot i for the 8 channels starting at c0 (wave-uniform) into (vo, mo)
    __device__ __forceinline__ void load_slot_to(float4 &vo, float4 &mo, const InSrc &s, int c0, int i, int clast) {
        const unsigned last = c0 == clast ? 0xffffffffu : 0u;           // (scalar)
        if (MODE == RAMNET_IN_S2D) {            // ld1 = log2(C0): parity group g = (a*2 + c) of the chunk -> pixel (2i+a, 2j+c)
            const int g = c0 >> s.ld1;
            vo = bload(r0, vo0[i] | (bad[i] & last), (((g >> 1) * 2 * s.Win + (g & 1)) * s.ld0 + (c0 - (g << s.ld1))) * 4);
        } else if (CAT) {
            // second (uniform; C0 % 8 == 0: a chunk lies in one tensor) selects descriptor and offsets — no branch, so that the
            // number of loads in flight is the same on every path (a join would force the compiler to drain them)
            const bool second = c0 >= s.C0;
            vo = bload(second ? r1 : r0, second ? (vo1[i] | (bad[i] & last)) : vo0[i], (second ? c0 - s.C0 : c0) * 4);
            if (MODE == RAMNET_IN_CAT_MUL) mo = bload(rm, second ? (vom[i] | (bad[i] & last)) : WOOB, (second ? c0 - s.C0 : 0) * 4);
        } else {
            vo = bload(r0, vo0[i] | (bad[i] & last), c0 * 4);
            if (MODE == RAMNET_IN_RELUMASK) mo = bload(rm, vom[i] | (bad[i] & last), c0 * 4);
        }
    }
    __device__ __forceinline__ void load_slot(const InSrc &s, int c0, int i, int clast) { load_slot_to(v[i], m[i], s, c0, i, clast); }
    __device__ __forceinline__ void load(const InSrc &s, int c0, int clast) {
#pragma unroll
        for (int i = 0; i < NS; ++i) load_slot(s, c0, i, clast);
    }
    // registers of slot i (loaded for channel c0) -> LDS patch
    // (r, mk) = what load_slot_to() left for slot i and channel c0 -> LDS patch
    __device__ __forceinline__ void store_regs(float *__restrict__ patch, const InSrc &s, int c0, int i, float4 r, float4 mk) const {
        if (MODE == RAMNET_IN_RELUMASK)
            r = make_float4(mk.x > 0.f ? r.x : 0.f, mk.y > 0.f ? r.y : 0.f, mk.z > 0.f ? r.z : 0.f, mk.w > 0.f ? r.w : 0.f);
        if (MODE == RAMNET_IN_CAT_MUL) {        // chunks of x0: the mask load returned zeros, scale by 1 instead (uniform select, no branch)
            const float one = c0 >= s.C0 ? 0.f : 1.f;
            r = make_float4(r.x * (mk.x + one), r.y * (mk.y + one), r.z * (mk.z + one), r.w * (mk.w + one));
        }
        st4(patch + ldst[i], r);
    }
    __device__ __forceinline__ void store_slot(float *__restrict__ patch, const InSrc &s, int c0, int i) const { store_regs(patch, s, c0, i, v[i], m[i]); }
    __device__ __forceinline__ void store(float *__restrict__ patch, const InSrc &s, int c0) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) store_slot(patch, s, c0, i);
    }
};

}  // namespace ramnet
