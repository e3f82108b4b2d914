// The two head layers of the RAM-Net path (5x5, stride 1, 5 event bins / 1 frame channel -> 32 feature maps at full
// resolution; statenet.py:160-175, submodules.py:8-35) for gfx950 (MI355X), forward and backward-weights.
//
// With 1-8 input channels the generic implicit-GEMM kernel has nothing to amortise its patch staging and operand traffic
// over (one 8-channel chunk, N = 32): it runs at ~16 % of the fp32 MFMA peak.  Here the reduction index is the dense
// (tap, channel) pair, K = 25*CR (125 for the event head instead of 25*8 = 200), and
//   forward:  the whole weight matrix [K][32] lives in REGISTERS (lane = output channel, 63 VGPRs for CR = 5); a wave owns
//             4 rows x 32 pixels, the A operand of v_mfma_f32_32x32x2_f32 is one 4-byte LDS read per MFMA from a
//             channel-planar input patch (conflict-free: lanes = consecutive pixels), addressed as lane base + a
//             compile-time (tap, channel) constant;
//   backward-weights: dW[(tap,c)][n] = sum_pixels in(pixel + tap, c) * g(pixel, n): M = (tap, c) in 4 blocks of 32, N = 32,
//             K = pixels; both operands from LDS (planar patch, gradient tile [pixel][32]), 5 reads per 4 MFMAs; the
//             [128][32] partial of a workgroup is folded into the OIHW gradient (and the bias gradient) by atomics.
// Exact fp32 like the rest of the path.
#include <stdlib.h>
#include "common.hpp"

namespace ramnet {

constexpr int HT_W = 32;                            // output pixels per workgroup: TH rows (TH/4 per wave) x 32
constexpr int HP_W = HT_W + 4;                      // input patch width
constexpr int HP_LD = HP_W + 1;                     // padded patch row (floats)
constexpr int HF_H = 16, HG_H = 8;                  // tile height of the forward / backward-weights kernel

template <int CR, int TH>
struct HeadGeom {
    static constexpr int KT = 25 * CR;              // dense reduction length
    static constexpr int NS = (KT + 1) / 2;         // MFMA steps (K = 2 each)
    static constexpr int PLANE = (TH + 4) * HP_LD;  // one channel plane of the input patch
    // LDS offset of reduction index k inside the planar patch, relative to the output pixel's own position
    static constexpr int koff(int k) {
        return k >= KT ? 0 : (k % CR) * PLANE + ((k / CR) / 5) * HP_LD + (k / CR) % 5;
    }
};

// stage the (TH+4 x HP_W) input patch of the tile at (b, oy0, ox0) as CR channel planes; zero outside the image
template <int CR, int TH>
__device__ __forceinline__ void head_stage_patch(float *__restrict__ P, const float *__restrict__ x, int ld, int b, int oy0, int ox0, int H,
                                                 int W, int tid) {
    constexpr int HP_PLANE = HeadGeom<CR, TH>::PLANE;
#pragma unroll
    for (int i = tid; i < (TH + 4) * HP_W; i += 256) {
        const int py = i / HP_W, px = i - py * HP_W;
        const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
        float v[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const float *src = x + ((size_t)(b * H + iy) * W + ix) * ld;
            const float4 a = ld4(src);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
            if (CR > 4) {
                const float4 c = ld4(src + 4);
                v[4] = c.x, v[5] = c.y, v[6] = c.z, v[7] = c.w;
            }
            if (CR > 8) {
                const float4 c = ld4(src + 8);
                v[8] = c.x, v[9] = c.y, v[10] = c.z, v[11] = c.w;
            }
        }
#pragma unroll
        for (int c = 0; c < CR; ++c) P[c * HP_PLANE + py * HP_LD + px] = v[c];
    }
}

static bool head_channels_ok(int c) { return c == 1 || c == 3 || c == 5 || c == 10; }      // (10: the configs[4] voxel grids)

struct HeadFwdParams {
    const float *x, *wp, *bias;
    float *out;
    int ld, ldo, B, H, W, Cout, relu, tiles_x, tiles_y;
};

template <int CR>
__global__ void __launch_bounds__(256, 2) conv_head_fwd_kernel(const HeadFwdParams p) {
    using G = HeadGeom<CR, HF_H>;
    constexpr int HT_H = HF_H;
    __shared__ float P[CR * G::PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, b = bid / p.tiles_y;
    const int oy0 = ty * HT_H, ox0 = tx * HT_W;

    // the lane's column of the weight matrix: k = 2s + h
    float wreg[G::NS];
#pragma unroll
    for (int s = 0; s < G::NS; ++s) wreg[s] = p.wp[(2 * s + h) * 32 + n];
    // (loaded HERE, in front of the patch staging that waits for all its loads: a bias load still pending at the epilogue makes the
    // compiler wait for ALL memory operations — i.e. for the previous STORE — in front of each of the 64 conditional stores)
    const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;

    head_stage_patch<CR, HF_H>(P, p.x, p.ld, b, oy0, ox0, p.H, p.W, tid);
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float *pa = P + (wave * 4) * HP_LD + n;      // A row = pixel (lane & 31) of patch row 4*wave + t (+ tap offsets)
#pragma unroll
    for (int s = 0; s < G::NS; ++s) {
        const int off = h ? G::koff(2 * s + 1) : G::koff(2 * s);
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = pa[off + t * HP_LD];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], wreg[s], acc[t], 0, 0, 0);
    }

    if (n >= p.Cout) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int oy = oy0 + wave * 4 + t;
        if (oy >= p.H) continue;
        float *orow = p.out + ((size_t)(b * p.H + oy) * p.W) * p.ldo + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ox >= p.W) continue;
            float v = acc[t][r] + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            orow[(size_t)ox * p.ldo] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward-weights
struct HeadWgradParams {
    const float *x, *g, *gm;
    float *dw, *dbias;
    int ld, ldg, ldgm, B, H, W, Cin, Cout, tiles_x, tiles_y, ntiles;
};

constexpr int HG_LD = 32;                           // gradient tile [HG_H * 32 pixels][32 channels]

template <int CR>
__global__ void __launch_bounds__(256, 2) conv_head_wgrad_kernel(const HeadWgradParams p) {
    using G = HeadGeom<CR, HG_H>;
    constexpr int HT_H = HG_H, HP_PLANE = G::PLANE, WR = HG_H / 4;     // WR rows of 32 pixels per wave
    constexpr int MT = (G::KT + 31) / 32;           // 32-row blocks of the (tap, channel) index
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *P = smem;                                // [CR][HG_H + 4][HP_LD]
    float *Gt = smem + ((CR * HP_PLANE + 3) & ~3);  // [HG_H * 32][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // A row m = (tap, channel) mt*32 + l31 at pixel 2j + h of the wave's WR x 32 strip; B row = the same pixel, column l31
    int aoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 32 + l31;
        const int tap = m / CR, c = m - tap * CR;
        aoff[t] = (m < G::KT ? c * HP_PLANE + (tap / 5) * HP_LD + tap % 5 : 0) + wave * WR * HP_LD + h;
    }
    const int boff = (wave * WR * 32 + h) * HG_LD + l31;
    float4 bsum = f4zero();                         // bias gradient partial of channel quad (tid & 7)

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int bid = tile;
        const int tx = bid % p.tiles_x;
        bid /= p.tiles_x;
        const int ty = bid % p.tiles_y, b = bid / p.tiles_y;
        const int oy0 = ty * HT_H, ox0 = tx * HT_W;
        __syncthreads();                            // the previous tile's readers are done
        head_stage_patch<CR, HG_H>(P, p.x, p.ld, b, oy0, ox0, p.H, p.W, tid);
#pragma unroll
        for (int i = 0; i < HT_H * HT_W * 8 / 256; ++i) {      // gradient tile: pixels x 8 channel quads
            const int sl = tid + i * 256, pix = sl >> 3, qd = sl & 7;
            const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
            float4 r = f4zero();
            if (oy < p.H && ox < p.W && qd * 4 < p.Cout) {
                const size_t gp = (size_t)(b * p.H + oy) * p.W + ox;
                r = ld4(p.g + gp * p.ldg + qd * 4);
                if (p.gm) {
                    const float4 mk = ld4(p.gm + gp * p.ldgm + qd * 4);
                    r = make_float4(mk.x > 0.f ? r.x : 0.f, mk.y > 0.f ? r.y : 0.f, mk.z > 0.f ? r.z : 0.f, mk.w > 0.f ? r.w : 0.f);
                }
            }
            st4(Gt + pix * HG_LD + qd * 4, r);
            bsum = f4add(bsum, r);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WR * 16; ++j) {         // the wave's pixels, two per MFMA
            const int po = ((2 * j) >> 5) * HP_LD + ((2 * j) & 31);
            const float bv = Gt[boff + 2 * j * HG_LD];
            float a[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = P[aoff[t] + po];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[t], 0, 0, 0);
        }
    }

    // fold the four waves' partials in LDS, then one atomic per element into ws[tap][Cin][Cout].  The waves take turns (plain
    // read-add-write: the lanes of ONE wave own distinct elements) — fp32 LDS atomics run at ~0.4 per clock and CU on gfx950
    // (profiles/r03_h_tuning_notes.md, voxelizer), MT * 4096 of them per workgroup were ~20 us of every workgroup's tail
    __syncthreads();
    float *red = smem;                              // [MT*32][32]  (red + bred = 5120 floats fit below the gradient tile's end)
    float *bred = smem + MT * 1024;                 // [32][32], behind red
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *q = red + (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31;
                    *q = (w == 0 ? 0.f : *q) + acc[t][r];
                }
        }
        if (w == 0) st4(bred + (tid >> 3) * 32 + (tid & 7) * 4, bsum);
        __syncthreads();
    }
    for (int i = tid; i < G::KT * 32; i += 256) {
        const int n = i & 31, m = i >> 5;
        const int tap = m / CR, c = m - tap * CR;
        if (n < p.Cout) atomicAdd(p.dw + ((size_t)tap * p.Cin + c) * p.Cout + n, red[i]);
    }
    if (p.dbias != nullptr && tid < 32 && tid < p.Cout) {
        float t = 0.f;
        for (int g = 0; g < 32; ++g) t += bred[g * 32 + tid];
        atomicAdd(p.dbias + tid, t);
    }
}

int launch_head_wgrad(const ramnet_wgrad_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 25 && d.stride == 1 && d.in_mode == RAMNET_IN_PLAIN);
    for (int t = 0; t < 25; ++t) RAMNET_CHECK_ARG(d.dy[t] == t / 5 - 2 && d.dx[t] == t % 5 - 2);
    RAMNET_CHECK_ARG(d.head_cin >= 1 && d.head_cin <= d.C0 && head_channels_ok(d.head_cin) && d.Cout <= 32);
    RAMNET_CHECK_ARG(d.Ho == d.Hin && d.Wo == d.Win && ((uintptr_t)d.x0 & 15) == 0 && ((uintptr_t)d.dout & 15) == 0 &&
                     (d.gmask == nullptr || ((uintptr_t)d.gmask & 15) == 0) && (d.head_cin <= 4 || d.C0 >= 8) && (d.head_cin <= 8 || d.C0 >= 12));
    HeadWgradParams q;
    q.x = d.x0, q.g = d.dout, q.gm = d.gmask, q.dw = d.dw, q.dbias = d.dbias;
    q.ld = d.ld0, q.ldg = d.ldg, q.ldgm = d.ldgm, q.B = d.B, q.H = d.Hin, q.W = d.Win, q.Cin = d.C0, q.Cout = d.Cout;
    q.tiles_x = cdiv(d.Wo, HT_W), q.tiles_y = cdiv(d.Ho, HG_H), q.ntiles = q.tiles_x * q.tiles_y * d.B;
    int blocks = 512;                               // persistent workgroups (256 measured the same)
    if (blocks > q.ntiles) blocks = q.ntiles;
    if (blocks < 1) blocks = 1;
    auto go = [&](auto kern, int cr) -> int {
        const size_t lds = (size_t)(((cr * (HG_H + 4) * HP_LD + 3) & ~3) + HG_H * HT_W * HG_LD) * sizeof(float);
        RAMNET_FULL_LDS((kern));
        note_kernel("conv_head_wgrad_kernel<%d>", cr);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, q);
        return 0;
    };
    int rc;
    if (d.head_cin == 1) rc = go(conv_head_wgrad_kernel<1>, 1);
    else if (d.head_cin == 3) rc = go(conv_head_wgrad_kernel<3>, 3);
    else if (d.head_cin == 10) rc = go(conv_head_wgrad_kernel<10>, 10);
    else rc = go(conv_head_wgrad_kernel<5>, 5);
    if (rc) return rc;
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// OIHW [Cout][Cin][5][5] -> [K pad][32]: row k = tap*Cin + c, zero rows / columns beyond
__global__ void pack_weight_head_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 32) return;
    const int n = i & 31, k = i >> 5;
    const int tap = k / Cin, c = k - tap * Cin;
    wp[i] = (tap < 25 && n < Cout) ? w[((size_t)n * Cin + c) * 25 + tap] : 0.f;
}

int launch_head(const ramnet_conv_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 25 && d.stride == 1 && d.in_mode == RAMNET_IN_PLAIN);
    for (int t = 0; t < 25; ++t) RAMNET_CHECK_ARG(d.dy[t] == t / 5 - 2 && d.dx[t] == t % 5 - 2 && d.wtap[t] == t);   // dense padded 5x5
    RAMNET_CHECK_ARG(d.head_cin >= 1 && d.head_cin <= d.C0 && head_channels_ok(d.head_cin) && d.Cout <= 32);
    RAMNET_CHECK_ARG(d.Ho == d.Hin && d.Wo == d.Win && d.HoF == d.Ho && d.WoF == d.Wo && d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0);
    RAMNET_CHECK_ARG((d.epi == RAMNET_EPI_RELU || d.epi == RAMNET_EPI_LINEAR) && d.beta == 0.f && d.frame == 0 && d.out_s2d == 0);
    RAMNET_CHECK_ARG(((uintptr_t)d.x0 & 15) == 0 && d.ld0 % 4 == 0 && (d.head_cin <= 4 || d.C0 >= 8) && (d.head_cin <= 8 || d.C0 >= 12));
    HeadFwdParams q;
    q.x = d.x0, q.wp = d.w, q.bias = d.bias, q.out = d.out;
    q.ld = d.ld0, q.ldo = d.ldo, q.B = d.B, q.H = d.Hin, q.W = d.Win, q.Cout = d.Cout, q.relu = d.epi == RAMNET_EPI_RELU;
    q.tiles_x = cdiv(d.Wo, HT_W), q.tiles_y = cdiv(d.Ho, HF_H);
    const dim3 grid(q.tiles_x * q.tiles_y * d.B);
    note_kernel("conv_head_fwd_kernel<%d>", d.head_cin == 1 ? 1 : d.head_cin == 3 ? 3 : 5);
    if (d.head_cin == 1) hipLaunchKernelGGL(conv_head_fwd_kernel<1>, grid, dim3(256), 0, st, q);
    else if (d.head_cin == 3) hipLaunchKernelGGL(conv_head_fwd_kernel<3>, grid, dim3(256), 0, st, q);
    else if (d.head_cin == 10) hipLaunchKernelGGL(conv_head_fwd_kernel<10>, grid, dim3(256), 0, st, q);
    else hipLaunchKernelGGL(conv_head_fwd_kernel<5>, grid, dim3(256), 0, st, q);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

extern "C" size_t ramnet_packed_weight_elems_head(int Cin) { return (size_t)((25 * Cin + 1) / 2 * 2) * 32; }

extern "C" int ramnet_head_supported(int Cin, int Cout) { return head_channels_ok(Cin) && Cout >= 1 && Cout <= 32; }

extern "C" int ramnet_pack_weight_head(const float *w, float *wp, int Cout, int Cin, void *stream) {
    RAMNET_CHECK_ARG(w && wp && ramnet_head_supported(Cin, Cout));
    const int rows = (25 * Cin + 1) / 2 * 2;
    hipLaunchKernelGGL(pack_weight_head_kernel, dim3(cdiv(rows * 32, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, rows);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
