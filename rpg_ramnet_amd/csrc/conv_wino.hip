// Winograd F(2x2, 3x3) convolution for gfx950 (MI355X): forward and backward-data of the 3x3 stride-1 layers of the
// RAM-Net path (ConvGRU gates / candidate, residual blocks), exact-fp32 arithmetic on v_mfma_f32_16x16x4_f32.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A         (Lavin & Gray 2016; 2.25x fewer multiplies than direct)
//
// A workgroup (4 waves) owns 8 x 16 output pixels = 4 x 8 Winograd tiles and 64 output channels.  For every chunk of 8
// input channels it stages the 10 x 18 input patch once (same fused loaders as the direct kernel: concatenation, the
// GRU's h*r product, the ReLU mask of the backward pass), transforms it to V[16 positions][32 tiles][8] in LDS, copies
// the pre-transformed weights U[16][64][8] (packed by ramnet_pack_weight_wino, LDS image = global image) and runs the 16
// independent [32 tiles x 8] x [8 x 64] products.  Each wave keeps ALL 16 positions of its 16 tiles x 32 channels in
// registers (16 x 2 accumulators of the 16x16 MFMA = 128 VGPRs), so the output transform A^T M A is register-local and
// the fused epilogues (bias / ReLU / sigmoid / residual / GRU blend) run straight from it.
// Global loads run one chunk ahead in registers; LDS fills and the input transform of chunk i+1 are issued under the
// MFMAs of chunk i (two barriers per chunk, see the pipeline comment in the kernel).
#include <stdlib.h>
#include "common.hpp"
#include "conv_epilogue.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ramnet {

constexpr int WK = 8;                         // input channels per chunk
constexpr int WBN = 64;                       // output channels per workgroup
constexpr int WTH = 8, WTW = 16;              // output pixels per workgroup (4 x 8 tiles of 2 x 2)
constexpr int WPH = WTH + 2, WPW = WTW + 2;   // input patch
constexpr int WU_FLOATS = 16 * WBN * WK;      // 32 KB
constexpr int WV_FLOATS = 16 * 32 * WK;       // 16 KB
constexpr int WP_FLOATS = WPH * WPW * WK;     // 5.6 KB

struct WinoParams {
    InSrc src;
    int nchunks, nblk;      // Cin/8, CoutPad/64
    int tiles_x, tiles_y;
    int dy0, dx0;           // offset of the first filter tap (-1 for the padded 3x3)
};

__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }

__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const ramnet_conv_desc p, const WinoParams q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *U = smem;                  // [16][64][8]  (row n: channel c at c ^ 4*((n>>3)&1))
    float *V = U + WU_FLOATS;         // [2][16][32][8]  (row t: same swizzle), double-buffered
    float *patch = V + 2 * WV_FLOATS; // [10][18][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, ks = lane >> 4;

    int bid = blockIdx.x;
    const int tx_i = bid % q.tiles_x;
    bid /= q.tiles_x;
    const int ty_i = bid % q.tiles_y;
    const int b = bid / q.tiles_y;
    const int n0 = blockIdx.y * WBN;
    const int oy0 = ty_i * WTH, ox0 = tx_i * WTW;
    const int iy0 = oy0 + q.dy0, ix0 = ox0 + q.dx0;

    // MFMA operand addresses: lane supplies row (l&15), k-slot (l>>4) = channels 2*ks, 2*ks+1 of the chunk
    const int swz = 4 * ((l15 >> 3) & 1);
    const int aoff = (wm * 16 + l15) * WK + ((ks * 2) ^ swz);
    const int boff = (wn * 32 + l15) * WK + ((ks * 2) ^ swz);

    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // input-transform role of this thread: row i of B^T d B for (tile, channel quad)
    const int ti = wave;                                 // 0..3 (wave-uniform)
    const int tq = tid & 1, tt = (tid >> 1) & 31;
    const int tty = tt >> 3, ttx = tt & 7;
    const int ra = ti == 0 ? 0 : (ti == 2 ? 2 : 1);     // rows combined: i0: d0-d2, i1: d1+d2, i2: d2-d1, i3: d1-d3
    const int rb = ti == 0 ? 2 : (ti == 1 ? 2 : (ti == 2 ? 1 : 3));
    const float sb = ti == 1 ? 1.f : -1.f;
    const float *pra = patch + ((2 * tty + ra) * WPW + 2 * ttx) * WK + tq * 4;
    const float *prb = patch + ((2 * tty + rb) * WPW + 2 * ttx) * WK + tq * 4;
    const int vdst = (ti * 4) * (32 * WK) + tt * WK + ((tq ^ ((tt >> 3) & 1)) * 4);

    PatchRegs<WK / 4, 2, 256> pr;
    float4 ulo[4], uhi[4];                         // weights of positions 0-7 / 8-15 in flight
    auto load_u = [&](float4 (&r)[4], int chunk, int half) {
        const float *src = p.w + ((size_t)chunk * q.nblk + blockIdx.y) * WU_FLOATS + half * (WU_FLOATS / 2) + tid * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = ld4(src + i * 1024);
    };
    auto store_u = [&](const float4 (&r)[4], int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(U + half * (WU_FLOATS / 2) + tid * 4 + i * 1024, r[i]);
    };
    auto transform = [&](float *vbuf) {            // patch -> row `ti` of B^T d B of (tile tt, channel quad tq)
        float4 e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 x = ld4(pra + c * WK), y = ld4(prb + c * WK);
            e[c] = make_float4(x.x + sb * y.x, x.y + sb * y.y, x.z + sb * y.z, x.w + sb * y.w);
        }
        const float4 v0 = make_float4(e[0].x - e[2].x, e[0].y - e[2].y, e[0].z - e[2].z, e[0].w - e[2].w);
        const float4 v1 = f4add(e[1], e[2]);
        const float4 v2 = make_float4(e[2].x - e[1].x, e[2].y - e[1].y, e[2].z - e[1].z, e[2].w - e[1].w);
        const float4 v3 = make_float4(e[1].x - e[3].x, e[1].y - e[3].y, e[1].z - e[3].z, e[1].w - e[3].w);
        st4(vbuf + vdst, v0);
        st4(vbuf + vdst + 32 * WK, v1);
        st4(vbuf + vdst + 2 * 32 * WK, v2);
        st4(vbuf + vdst + 3 * 32 * WK, v3);
    };
    auto mma = [&](const float *vbuf, int pos0) {  // 8 of the 16 positions
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int pos = pos0 + pp;
            const float2 a = ld2(vbuf + pos * (32 * WK) + aoff);
            const float2 b0 = ld2(U + pos * (WBN * WK) + boff);
            const float2 b1 = ld2(U + pos * (WBN * WK) + boff + 16 * WK);
            acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[pos][0], 0, 0, 0);
            acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[pos][1], 0, 0, 0);
            acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[pos][0], 0, 0, 0);
            acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[pos][1], 0, 0, 0);
        }
    };

    // Two-phase software pipeline, two barriers per chunk, every LDS fill under the MFMAs of the other half:
    //   phase 1 (positions 0-7 of chunk i):  U_hi(i) and patch(i+1) are written   (their readers finished at barrier X)
    //   phase 2 (positions 8-15 of chunk i): patch(i+1) -> V[(i+1)&1], U_lo(i+1) written  (readers finished at barrier Y)
    // Global loads are issued one full iteration before the registers are stored to LDS.
    const int nch = q.nchunks;
    pr.load(q.src, b, iy0, ix0, 0, WPH, WPW, tid);
    load_u(ulo, 0, 0);
    load_u(uhi, 0, 1);
    pr.store<WK>(patch, q.src, WPH, WPW, tid);
    __syncthreads();
    transform(V);
    store_u(ulo, 0);
    if (nch > 1) {
        pr.load(q.src, b, iy0, ix0, WK, WPH, WPW, tid);
        load_u(ulo, 1, 0);
    }
    __syncthreads();                               // barrier X_0
    for (int chunk = 0; chunk < nch; ++chunk) {
        const float *vcur = V + (chunk & 1) * WV_FLOATS;
        float *vnext = V + ((chunk + 1) & 1) * WV_FLOATS;
        store_u(uhi, 1);
        if (chunk + 1 < nch) {
            pr.store<WK>(patch, q.src, WPH, WPW, tid);
            load_u(uhi, chunk + 1, 1);
            if (chunk + 2 < nch) pr.load(q.src, b, iy0, ix0, (chunk + 2) * WK, WPH, WPW, tid);
        }
        mma(vcur, 0);
        __syncthreads();                           // barrier Y: U_hi(i), patch(i+1) visible; U_lo(i) free
        if (chunk + 1 < nch) {
            transform(vnext);
            store_u(ulo, 0);
            if (chunk + 2 < nch) load_u(ulo, chunk + 2, 0);
        }
        mma(vcur, 8);
        __syncthreads();                           // barrier X: V(i+1), U_lo(i+1) visible; U_hi(i), V(i), patch free
    }

    // ---- output transform + epilogue.  D of the 16x16 MFMA: col = lane&15 (channel), row = 4*(lane>>4) + r (tile)
    const int epi = p.epi;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = n0 + wn * 32 + f * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[i * 4 + 0][f][r], m1 = acc[i * 4 + 1][f][r], m2 = acc[i * 4 + 2][f][r], m3 = acc[i * 4 + 3][f][r];
                t[i][0] = m0 + m1 + m2;
                t[i][1] = m1 - m2 - m3;
            }
            float y[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                y[0][c] = t[0][c] + t[1][c] + t[2][c];
                y[1][c] = t[1][c] - t[2][c] - t[3][c];
            }
            const int tile = wm * 16 + 4 * ks + r;
            const int ty = tile >> 3, tx = tile & 7;
            if (n >= p.Cout) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int oy = oy0 + 2 * ty + a, ox = ox0 + 2 * tx + c;
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    const size_t pix = ((size_t)b * p.HoF + (oy * p.osy + p.ooy)) * p.WoF + (ox * p.osx + p.oox);
                    epilogue_store(p, epi, pix, n, y[a][c]);
                }
        }
    }
}

// OIHW 3x3 -> U = G g G^T in the kernel's LDS image [chunk8][block64][pos 16][64][8 swizzled]; evaluated in double.
__global__ void pack_weight_wino_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int transposed,
                                        int R, int N, int nchunks, int nblk, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cphys = (int)(i % WK);
        size_t j = i / WK;
        const int n = (int)(j % WBN);
        j /= WBN;
        const int pos = (int)(j % 16);
        j /= 16;
        const int nb = (int)(j % nblk);
        const int chunk = (int)(j / nblk);
        const int c = cphys ^ (4 * ((n >> 3) & 1));
        const int r = chunk * WK + c, no = nb * WBN + n;
        float v = 0.f;
        if (r < R && no < N) {
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb)
                    g[a][bb] = transposed ? (double)w[((size_t)r * Cin + no) * 9 + (2 - a) * 3 + (2 - bb)]
                                          : (double)w[((size_t)no * Cin + r) * 9 + a * 3 + bb];
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const int pi = pos >> 2, pj = pos & 3;
            double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) s += G[pi][a] * g[a][bb] * G[pj][bb];
            v = (float)s;
        }
        wp[i] = v;
    }
}

static void wino_geometry(int Cout, int Cin, int transposed, int &R, int &N, int &nchunks, int &nblk) {
    R = transposed ? Cout : Cin;
    N = transposed ? Cin : Cout;
    nchunks = cdiv(R, WK);
    nblk = cdiv(N, WBN);
}

int launch_wino(const ramnet_conv_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 9 && d.stride == 1 && d.precision == RAMNET_PREC_F32 && d.epi != RAMNET_EPI_LSTM);
    RAMNET_CHECK_ARG(d.in_mode != RAMNET_IN_UP2X && d.in_mode != RAMNET_IN_UP2X_SKIP);
    // the taps must be the dense 3x3 window; which weight slice each one reads is baked into the Winograd pack
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
    q.nchunks = cdiv(q.src.Cin, WK), q.nblk = cdiv(d.Cout, WBN);
    q.tiles_x = cdiv(d.Wo, WTW), q.tiles_y = cdiv(d.Ho, WTH);
    q.dy0 = dymin, q.dx0 = dxmin;
    const size_t lds = (size_t)(WU_FLOATS + 2 * WV_FLOATS + WP_FLOATS) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        RAMNET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid(q.tiles_x * q.tiles_y * d.B, q.nblk);
    hipLaunchKernelGGL(conv_wino_kernel, grid, dim3(256), lds, st, d, q);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

extern "C" size_t ramnet_packed_weight_elems_wino(int Cout, int Cin, int transposed) {
    int R, N, nchunks, nblk;
    wino_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    return (size_t)nchunks * nblk * WU_FLOATS;
}

extern "C" int ramnet_pack_weight_wino(const float *w, float *wp, int Cout, int Cin, int transposed, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0);
    int R, N, nchunks, nblk;
    wino_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    const size_t total = (size_t)nchunks * nblk * WU_FLOATS;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, transposed,
                       R, N, nchunks, nblk, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
