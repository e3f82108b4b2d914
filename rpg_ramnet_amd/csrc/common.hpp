// Shared device/host helpers for the gfx950 RAM-Net kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ramnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace ramnet {

constexpr int CK = 16;        // input channels staged per chunk by the forward / backward-data kernel
constexpr int LDP = CK + 4;   // padded LDS row (floats) of a patch pixel / weight row: conflict-free b128 reads
constexpr int TWID = 16;      // output tile width in pixels (tile = TH x 16)
constexpr int WCK = 32;       // input channels per block of the weight-gradient kernel

extern int g_opt_voxel_sorted, g_opt_fold_pair, g_opt_wgrad_blocks, g_opt_wgrad_wino_blocks, g_opt_wino_ksplit, g_opt_wgrad_wino_nf, g_opt_pred_si_cap, g_opt_pred_si_bwd_cap;     // ramnet_set_option() (pointwise.hip)
void set_error(const char *fmt, ...);
void note_kernel(const char *fmt, ...);   // symbol (template arguments included) of the MFMA kernel a launcher enqueued: ramnet_last_kernel()
int launch_wino(const ramnet_conv_desc &d, hipStream_t st);   // conv_wino.hip
size_t wino24_splitk_floats(const ramnet_conv_desc &d);      // conv_wino24.hip
int launch_wino6(const ramnet_conv_desc &d, hipStream_t st);  // conv_wino6.hip: F(2x4,3x3)
int launch_wino6s(const ramnet_conv_desc &d, hipStream_t st); // conv_wino6s.hip: F(2x4,3x3), split bf16 operands
int launch_head(const ramnet_conv_desc &d, hipStream_t st);   // conv_head.hip
int launch_wino24(const ramnet_conv_desc &d, hipStream_t st); // conv_wino24.hip
int launch_wgrad_wino24(const ramnet_wgrad_desc &d, hipStream_t st);   // conv_wgrad_wino24.hip
int launch_head_wgrad(const ramnet_wgrad_desc &d, hipStream_t st);
int launch_wgrad_wino(const ramnet_wgrad_desc &d, hipStream_t st);   // conv_wgrad_wino.hip
int launch_wgrad_wino6(const ramnet_wgrad_desc &d, hipStream_t st);  // conv_wgrad_wino6.hip: F(2x4,3x3)
int launch_wgrad_dsplit(const ramnet_wgrad_desc &d, hipStream_t st); // conv_wgrad_dsplit.hip: direct 3x3, split bf16 operands

#define RAMNET_CHECK_ARG(cond)                                                        \
    do {                                                                              \
        if (!(cond)) {                                                                \
            ramnet::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);  \
            return RAMNET_E_BADARG;                                                   \
        }                                                                             \
    } while (0)

#define RAMNET_HIP(call)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ramnet::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return (int)e_;                                                                     \
        }                                                                                       \
    } while (0)

#define RAMNET_LAUNCH_CHECK() RAMNET_HIP(hipGetLastError())

// Raise a kernel's dynamic-LDS cap to the 160 KB of a CU — once per kernel and process, not per launch (a driver call on the
// launch path, and nothing that belongs inside a hipGraph stream capture).
hipError_t allow_full_lds(const void *kernel);
#define RAMNET_FULL_LDS(kern) RAMNET_HIP(ramnet::allow_full_lds(reinterpret_cast<const void *>(kern)))

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, 1 ulp each): absolute error <= 3e-7, far
// inside the 2e-4 parity bar, and ~5 instructions instead of ~25 for expf + IEEE division — the conv epilogues evaluate 32 of
// them per thread between the last MFMA and the stores.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Buffer resource over [p, p + bytes): loads take a 32-bit per-lane byte offset plus a scalar one, and an offset past `bytes`
// returns 0 — the zero padding of the convolution costs no instruction.  The pointer goes through readfirstlane so that the
// compiler knows the descriptor is wave-uniform (cdna_hip_programming.md, buffer addressing).
__device__ __forceinline__ auto wino_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
}
constexpr unsigned WOOB = 0x7fffff00u;          // byte offset past every image (images are < 2 GB: checked on the host)

// Everything the patch loader needs to know about the logical input of a convolution.
struct InSrc {
    const float *x0, *x1, *xm;
    int ld0, ld1, ldm, C0, Cin, mode, Hin, Win;
};

// Source row/col and blend weight of one output coordinate of F.interpolate(scale_factor=2,
// mode='bilinear', align_corners=False): src = max(0.5*(dst+0.5)-0.5, 0).
__device__ __forceinline__ void up2x_coord(int d, int n_src, int &i0, int &i1, float &l1) {
    float s = fmaxf(0.5f * (float)d - 0.25f, 0.0f);
    i0 = (int)s;
    i1 = i0 + (i0 < n_src - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// Four consecutive input channels c..c+3 of logical input pixel (b, iy, ix); zero outside the image
// (the convolution's zero padding) and beyond the last channel (chunk padding).
__device__ __forceinline__ float4 load_in4(const InSrc &s, int b, int iy, int ix, int c) {
    if ((unsigned)iy >= (unsigned)s.Hin || (unsigned)ix >= (unsigned)s.Win || c >= s.Cin) return f4zero();
    if (s.mode == RAMNET_IN_UP2X || s.mode == RAMNET_IN_UP2X_SKIP) {
        const int Hs = s.Hin >> 1, Ws = s.Win >> 1;
        int y0, y1, x0, x1;
        float ly, lx;
        up2x_coord(iy, Hs, y0, y1, ly);
        up2x_coord(ix, Ws, x0, x1, lx);
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const size_t r0 = ((size_t)b * Hs + y0) * Ws, r1 = ((size_t)b * Hs + y1) * Ws;
        float4 v00 = ld4(s.x0 + (r0 + x0) * s.ld0 + c), v01 = ld4(s.x0 + (r0 + x1) * s.ld0 + c);
        float4 v10 = ld4(s.x0 + (r1 + x0) * s.ld0 + c), v11 = ld4(s.x0 + (r1 + x1) * s.ld0 + c);
        if (s.mode == RAMNET_IN_UP2X_SKIP) {
            v00 = f4add(v00, ld4(s.x1 + (r0 + x0) * s.ld1 + c));
            v01 = f4add(v01, ld4(s.x1 + (r0 + x1) * s.ld1 + c));
            v10 = f4add(v10, ld4(s.x1 + (r1 + x0) * s.ld1 + c));
            v11 = f4add(v11, ld4(s.x1 + (r1 + x1) * s.ld1 + c));
        }
        float4 top = f4add(f4scale(v00, hx), f4scale(v01, lx));
        float4 bot = f4add(f4scale(v10, hx), f4scale(v11, lx));
        return f4add(f4scale(top, hy), f4scale(bot, ly));
    }
    const size_t pix = ((size_t)b * s.Hin + iy) * s.Win + ix;
    if (s.mode == RAMNET_IN_PLAIN) return ld4(s.x0 + pix * s.ld0 + c);
    if (s.mode == RAMNET_IN_RELUMASK) {
        float4 v = ld4(s.x0 + pix * s.ld0 + c), m = ld4(s.xm + pix * s.ldm + c);
        return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
    }
    // CAT / CAT_MUL
    if (c < s.C0) return ld4(s.x0 + pix * s.ld0 + c);
    float4 v = ld4(s.x1 + pix * s.ld1 + (c - s.C0));
    if (s.mode == RAMNET_IN_CAT_MUL) v = f4mul(v, ld4(s.xm + pix * s.ldm + (c - s.C0)));
    return v;
}

// ---- batched, branch-free patch staging ---------------------------------------------------------------------
// Stages PH x PW pixels x (4*QPP) channels starting at channel c0 into LDS rows of LDX floats.  Slots are handled in
// batches of NB: all global loads of a batch are issued before the first LDS store, so a thread exposes ONE memory
// latency per batch instead of one per float4 (the naive loop serialises on every load).  Out-of-image pixels and
// channels >= Cin read a safe address and are zeroed by a select (no divergent branches around the loads).
template <int QPP, int LDX, int NB, int NT>
__device__ __forceinline__ void stage_patch(float *__restrict__ patch, const InSrc &s, int b, int iy0, int ix0, int c0,
                                            int PH, int PW, int tid) {
    constexpr int PXS = LDX, QS = 4;               // per-pixel / per-quad stride (floats)
    const int nslots = PH * PW * QPP;
    if (s.mode == RAMNET_IN_UP2X || s.mode == RAMNET_IN_UP2X_SKIP) {   // 4-8 loads per slot already in flight
        for (int sl = tid; sl < nslots; sl += NT) {
            const int pix = sl / QPP, qd = sl - pix * QPP;
            const int py = pix / PW, px = pix - py * PW;
            st4(patch + pix * PXS + qd * QS, load_in4(s, b, iy0 + py, ix0 + px, c0 + qd * 4));
        }
        return;
    }
    const bool two = s.mode == RAMNET_IN_CAT_MUL || s.mode == RAMNET_IN_RELUMASK;
    for (int base = tid; base < nslots; base += NB * NT) {
        float4 v[NB], m[NB];
        int dst[NB];
        bool ok[NB], hasm[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = base + i * NT;
            const int pix = sl / QPP, qd = sl - pix * QPP;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px, c = c0 + qd * 4;
            ok[i] = sl < nslots && (unsigned)iy < (unsigned)s.Hin && (unsigned)ix < (unsigned)s.Win && c < s.Cin;
            dst[i] = sl < nslots ? pix * PXS + qd * QS : -1;
            const size_t gp = ((size_t)b * s.Hin + iy) * s.Win + ix;
            const bool second = s.mode != RAMNET_IN_PLAIN && s.mode != RAMNET_IN_RELUMASK && c >= s.C0;
            const float *p0 = second ? s.x1 + gp * s.ld1 + (c - s.C0) : s.x0 + gp * s.ld0 + c;
            hasm[i] = two && (s.mode == RAMNET_IN_RELUMASK || second);
            const float *p1 = s.mode == RAMNET_IN_RELUMASK ? s.xm + gp * s.ldm + c : s.xm + gp * s.ldm + (c - s.C0);
            v[i] = ld4(ok[i] ? p0 : s.x0);
            if (two) m[i] = ld4(ok[i] && hasm[i] ? p1 : s.x0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float4 r = v[i];
            if (two && hasm[i]) {
                if (s.mode == RAMNET_IN_RELUMASK)
                    r = make_float4(m[i].x > 0.f ? r.x : 0.f, m[i].y > 0.f ? r.y : 0.f, m[i].z > 0.f ? r.z : 0.f, m[i].w > 0.f ? r.w : 0.f);
                else
                    r = f4mul(r, m[i]);
            }
            if (!ok[i]) r = f4zero();
            if (dst[i] >= 0) st4(patch + dst[i], r);
        }
    }
}

// The same staging split in two halves so that the global loads of the NEXT chunk can be kept in flight (in registers)
// under the MFMAs of the current one: load() issues them, store() applies mask / product / zero padding and writes LDS.
// (PLAIN / CAT / CAT_MUL / RELUMASK sources; the upsampling loaders are not needed by the 3x3 layers that use it.)
template <int QPP, int NB, int NT>
struct PatchRegs {
    float4 v[NB], m[NB];
    unsigned okmask, mmask;

    __device__ __forceinline__ void load(const InSrc &s, int b, int iy0, int ix0, int c0, int PH, int PW, int tid) {
        const int nslots = PH * PW * QPP;
        const bool two = s.mode == RAMNET_IN_CAT_MUL || s.mode == RAMNET_IN_RELUMASK;
        okmask = 0, mmask = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = tid + i * NT;
            const int pix = sl / QPP, qd = sl - pix * QPP;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px, c = c0 + qd * 4;
            const bool ok = sl < nslots && (unsigned)iy < (unsigned)s.Hin && (unsigned)ix < (unsigned)s.Win && c < s.Cin;
            const size_t gp = ((size_t)b * s.Hin + iy) * s.Win + ix;
            const bool second = s.mode != RAMNET_IN_PLAIN && s.mode != RAMNET_IN_RELUMASK && c >= s.C0;
            const float *p0 = second ? s.x1 + gp * s.ld1 + (c - s.C0) : s.x0 + gp * s.ld0 + c;
            const bool hm = two && (s.mode == RAMNET_IN_RELUMASK || second);
            const float *p1 = s.mode == RAMNET_IN_RELUMASK ? s.xm + gp * s.ldm + c : s.xm + gp * s.ldm + (c - s.C0);
            v[i] = ld4(ok ? p0 : s.x0);
            m[i] = two ? ld4(ok && hm ? p1 : s.x0) : f4zero();
            okmask |= (ok ? 1u : 0u) << i;
            mmask |= (hm ? 1u : 0u) << i;
        }
    }

    // value of slot i after mask / product / zero padding
    __device__ __forceinline__ float4 value(const InSrc &s, int i) const {
        float4 r = v[i];
        if ((mmask >> i) & 1u) {
            if (s.mode == RAMNET_IN_RELUMASK)
                r = make_float4(m[i].x > 0.f ? r.x : 0.f, m[i].y > 0.f ? r.y : 0.f, m[i].z > 0.f ? r.z : 0.f, m[i].w > 0.f ? r.w : 0.f);
            else
                r = f4mul(r, m[i]);
        }
        if (!((okmask >> i) & 1u)) r = f4zero();
        return r;
    }

    // channel-pair planes: patch[pair][row][ROWF] (2 floats per pixel), planes PS floats apart — the layout whose
    // b64 reads by (tile, channel pair) lanes are bank-conflict free (conv_wino.hip)
    template <int ROWF, int PS>
    __device__ __forceinline__ void store_planes(float *__restrict__ patch, const InSrc &s, int PH, int PW, int tid) const {
        const int nslots = PH * PW * QPP;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = tid + i * NT;
            const float4 r = value(s, i);
            const int pix = sl / QPP, qd = sl - pix * QPP;
            const int py = pix / PW, px = pix - py * PW;
            if (sl < nslots) {
                float *d = patch + (2 * qd) * PS + py * ROWF + px * 2;
                *reinterpret_cast<float2 *>(d) = make_float2(r.x, r.y);
                *reinterpret_cast<float2 *>(d + PS) = make_float2(r.z, r.w);
            }
        }
    }

    template <int LDX>
    __device__ __forceinline__ void store(float *__restrict__ patch, const InSrc &s, int PH, int PW, int tid) const {
        const int nslots = PH * PW * QPP;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = tid + i * NT;
            float4 r = v[i];
            if ((mmask >> i) & 1u) {
                if (s.mode == RAMNET_IN_RELUMASK)
                    r = make_float4(m[i].x > 0.f ? r.x : 0.f, m[i].y > 0.f ? r.y : 0.f, m[i].z > 0.f ? r.z : 0.f, m[i].w > 0.f ? r.w : 0.f);
                else
                    r = f4mul(r, m[i]);
            }
            if (!((okmask >> i) & 1u)) r = f4zero();
            if (sl < nslots) st4(patch + (sl / QPP) * LDX + (sl % QPP) * 4, r);
        }
    }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int roundup(int a, int b) { return cdiv(a, b) * b; }

}  // namespace ramnet
