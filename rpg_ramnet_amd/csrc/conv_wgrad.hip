// Backward-weights (+bias) of every convolution on the RAM-Net path, gfx950 fp32 MFMA.
//
//   dW[t][c][n] += sum_{b,oy,ox} in(b, oy*s+dy[t], ox*s+dx[t], c) * g(b, oy, ox, n)
//
// GEMM view: M = input channel (32 per workgroup), N = output channel (64 per workgroup), K = pixels.
// A workgroup walks TH x 16 pixel tiles; per tile it stages the input patch (tile + halo, 32
// channels; same fused loaders as the forward kernel: concat, h*r, bilinear x2 (+skip), ReLU mask)
// and the 128 x 64 gradient tile in LDS, then every filter tap is one 32x32 accumulator whose A
// operand is the patch shifted by the tap offset.  The taps x 2 N-halves accumulators are spread
// over the 4 waves (wave w: N-half w&1, taps (w>>1), (w>>1)+2, ...) and stay in registers for the
// whole pixel range of the workgroup; partial sums of the pixel splits meet in a [tap][Cin][Cout]
// fp32 workspace through coalesced atomic adds.  The bias gradient (column sums of g) rides along.
#include <stdlib.h>
#include "common.hpp"

namespace ramnet {


struct WgradDerived {
    InSrc src;
    int PH, PW, dymin, dxmin;
    int TH;                         // pixel tile = TH x 16: 8, or 4 when two workgroups would not fit a CU's LDS (stride-2 5x5)
    int tiles_x, tiles_y, ntiles;   // pixel tiles per image / total
    int tpm, ntt;                   // taps per 32-row accumulator tile (Cin < 32 packs several taps), number of tap tiles
    int toff[25];
    int gsy, gsx, goy, gox, HoG, WoG;   // addressing of dout / gmask (ramnet_wgrad_desc)
};

// MAXT: accumulator tiles per wave; NSUB: 32-wide output-channel sub-tiles per workgroup (WBN = 32*NSUB).
// Accumulator tile (jt, ns): rows = (tap jt*tpm + row/cinp, channel row%cinp), cols = output channel ns*32 + col.
// Wave w owns ns = w % NSUB and tap tiles w/NSUB, w/NSUB + 4/NSUB, ...; waves with fewer real tiles run the same MAXT
// MFMAs on a valid dummy address (they would wait at the barrier anyway) so the K loop has no branches.
// CSUB = 2 (3x3 layers with >= 64 input channels): the workgroup owns 64 input x 64 output channels and wave w owns ALL taps
// of quadrant (input half w>>1, output half w&1): 9 tiles per wave, perfectly balanced, and the gradient tile is staged
// once per 64 input channels instead of once per 32.
template <int MAXT, int NSUB, int CSUB = 1>
__global__ void __launch_bounds__(256, CSUB == 2 ? 2 : 1) conv_wgrad_kernel(const ramnet_wgrad_desc p, const WgradDerived q) {
    constexpr int WBN = 32 * NSUB, GQ = WBN / 4;   // GQ: channel quads per gradient-tile pixel
    constexpr int WCK = 32 * CSUB;                 // input channels per workgroup (shadows the namespace default)
    constexpr int TSTEP = CSUB == 2 ? 1 : 4 / NSUB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                         // [PH*PW][32]
    float *gsm = smem + q.PH * q.PW * WCK;       // [128][WBN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int ns = wave % NSUB, csub = CSUB == 2 ? wave / NSUB : 0, tap0 = CSUB == 2 ? 0 : wave / NSUB;
    const int ntw = (q.ntt - tap0 + TSTEP - 1) / TSTEP;
    const int c0 = blockIdx.y * WCK, n0 = blockIdx.z * WBN;
    const int cinp = 32 / q.tpm, sub = l31 / cinp, cl = l31 - sub * cinp;

    f32x16 acc[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int toffw[MAXT];                             // per-lane LDS offset of this lane's (tap, channel) row
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int tap = (tap0 + TSTEP * j) * q.tpm + sub;
        toffw[j] = ((j < ntw && tap < p.ntaps) ? q.toff[tap] + cl : cl) + csub * 32;
    }

    float4 bsum = f4zero();                      // bias gradient partial: fixed channel quad (tid % GQ)
    const bool do_bias = p.dbias != nullptr && blockIdx.y == 0;

    for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        int tt = tile;
        const int tx_i = tt % q.tiles_x;
        tt /= q.tiles_x;
        const int ty_i = tt % q.tiles_y;
        const int b = tt / q.tiles_y;
        const int oy0 = ty_i * q.TH, ox0 = tx_i * TWID;
        const int iy0 = oy0 * p.stride + q.dymin, ix0 = ox0 * p.stride + q.dxmin;
        __syncthreads();
        {   // gradient tile: TH*16 pixels x GQ channel quads, 4 loads (x2 with mask) in flight per batch
            constexpr int NG = 128 * GQ / 256;
            const int nsl = q.TH * TWID * GQ;
            stage_patch<WCK / 4, WCK, 4, 256>(patch, q.src, b, iy0, ix0, c0, q.PH, q.PW, tid);
#pragma unroll
            for (int h = 0; h < NG; h += 4) {
                float4 g[4], y[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = tid + (h + i) * 256;
                    const int m = s / GQ, qd = s % GQ;
                    const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15), n = n0 + qd * 4;
                    ok[i] = s < nsl && oy < p.Ho && ox < p.Wo && n < p.Cout;
                    const size_t pix = ((size_t)b * q.HoG + (oy * q.gsy + q.goy)) * q.WoG + (ox * q.gsx + q.gox);
                    g[i] = ld4(ok[i] ? p.dout + pix * p.ldg + n : p.dout);
                    if (p.gmask) y[i] = ld4(ok[i] ? p.gmask + pix * p.ldgm + n : p.gmask);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = tid + (h + i) * 256;
                    float4 r = g[i];
                    if (p.gmask) r = make_float4(y[i].x > 0.f ? r.x : 0.f, y[i].y > 0.f ? r.y : 0.f, y[i].z > 0.f ? r.z : 0.f, y[i].w > 0.f ? r.w : 0.f);
                    if (!ok[i]) r = f4zero();
                    if (s < nsl) st4(gsm + (s / GQ) * WBN + (s % GQ) * 4, r);
                    bsum = f4add(bsum, r);
                }
            }
        }
        __syncthreads();
        {   // K step = 2 pixels (lanes 0-31: pixel 2i, lanes 32-63: pixel 2i+1).  Two-stage operand pipeline pinned with
            // sched_barrier: the LDS reads of step i+1 are issued BEFORE the MFMAs of step i (left alone, hipcc sinks the
            // reads next to their use and every step eats a full LDS latency whenever the wave is alone on its SIMD).
            float a0[MAXT], a1[MAXT], b0, b1;
            auto fetch = [&](int i, float (&a)[MAXT], float &bv) {
                const int m = 2 * i + kk;
                bv = gsm[m * WBN + ns * 32 + l31];
                const float *pa = patch + (((m >> 4) * p.stride) * q.PW + (m & 15) * p.stride) * WCK;
#pragma unroll
                for (int j = 0; j < MAXT; ++j) a[j] = pa[toffw[j]];
            };
            auto mma = [&](const float (&a)[MAXT], float bv) {
#pragma unroll
                for (int j = 0; j < MAXT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv, acc[j], 0, 0, 0);
            };
            fetch(0, a0, b0);
            const int ksteps = q.TH * TWID / 2;
            for (int i = 0; i < ksteps; i += 2) {
                fetch(i + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(i + 2 < ksteps ? i + 2 : i, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // D[row = (tap, input channel)][col = output channel] -> ws[(tap*Cin + c)*Cout + n]
    const int Cin = q.src.Cin;
    const int n = n0 + ns * 32 + l31;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        if (j >= ntw) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int t = (tap0 + TSTEP * j) * q.tpm + row / cinp, c = c0 + csub * 32 + row % cinp;
            if (t < p.ntaps && c < Cin && n < p.Cout) atomicAdd(p.dw + ((size_t)t * Cin + c) * p.Cout + n, acc[j][r]);
        }
    }
    if (do_bias) {
        __syncthreads();
        float *red = smem;                        // [256 / GQ][WBN]
        st4(red + (tid / GQ) * WBN + (tid % GQ) * 4, bsum);
        __syncthreads();
        if (tid < WBN) {
            float s = 0.f;
            for (int g = 0; g < 256 / GQ; ++g) s += red[g * WBN + tid];
            if (n0 + tid < p.Cout) atomicAdd(p.dbias + n0 + tid, s);
        }
    }
}

template <int MAXT, int NSUB, int CSUB = 1>
static int launch_wgrad(const ramnet_wgrad_desc &d, const WgradDerived &q, hipStream_t st) {
    auto kern = conv_wgrad_kernel<MAXT, NSUB, CSUB>;
    constexpr int WBN = 32 * NSUB, WCK = 32 * CSUB;
    size_t lds = ((size_t)q.PH * q.PW * WCK + q.TH * TWID * WBN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        RAMNET_FULL_LDS((kern));
        attr_set = true;
    }
    if (lds > 160 * 1024) {
        set_error("wgrad patch does not fit LDS (%zu bytes)", lds);
        return RAMNET_E_UNSUPPORTED;
    }
    const int gy = cdiv(q.src.Cin, WCK), gz = cdiv(d.Cout, WBN);
    // one resident wave of workgroups (256 CUs x blocks/CU the register budget admits): fewer pixel splits = fewer
    // atomic partial-sum merges, and every CU still gets an equal share
    int splits = g_opt_wgrad_blocks / (gy * gz);                // (ramnet_set_option("wgrad_blocks"): default 512)
    if (splits > q.ntiles) splits = q.ntiles;
    if (splits < 1) splits = 1;
    note_kernel("conv_wgrad_kernel<%d,%d,%d>", MAXT, NSUB, CSUB);
    hipLaunchKernelGGL(kern, dim3(splits, gy, gz), dim3(256), lds, st, d, q);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_wgrad_launch(const ramnet_wgrad_desc *dp, void *stream) {
    RAMNET_CHECK_ARG(dp != nullptr);
    const ramnet_wgrad_desc &d = *dp;
    RAMNET_CHECK_ARG(d.x0 && d.dout && d.dw);
    RAMNET_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 25 && (d.stride == 1 || d.stride == 2));
    RAMNET_CHECK_ARG(d.B > 0 && d.Ho > 0 && d.Wo > 0 && d.Cout > 0 && d.Cout % 4 == 0 && d.ldg % 4 == 0);
    RAMNET_CHECK_ARG(d.C0 > 0 && d.C0 % 4 == 0 && d.ld0 % 4 == 0);
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    if (cat) RAMNET_CHECK_ARG(d.x1 && d.C1 > 0 && d.C1 % 4 == 0 && d.ld1 % 4 == 0);
    if (d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK) RAMNET_CHECK_ARG(d.xm && d.ldm % 4 == 0);
    if (d.in_mode == RAMNET_IN_UP2X_SKIP) RAMNET_CHECK_ARG(d.x1 && d.ld1 % 4 == 0);
    if (d.gmask) RAMNET_CHECK_ARG(d.ldgm % 4 == 0);
    RAMNET_CHECK_ARG(d.nseg >= 0 && (d.nseg == 0 || d.algo == RAMNET_ALGO_WINOGRAD_2X4));       // multi-segment launches: F(2x4,3x3) only
    if (d.algo == RAMNET_ALGO_WINOGRAD24) return launch_wgrad_wino24(d, (hipStream_t)stream);
    const bool gdense = d.gsy == 0 && d.gsx == 0 && d.goy == 0 && d.gox == 0 && d.HoG == 0 && d.WoG == 0;
    if (!gdense) RAMNET_CHECK_ARG(d.gsy >= 1 && d.gsx >= 1 && d.goy >= 0 && d.gox >= 0 && (d.Ho - 1) * d.gsy + d.goy < d.HoG &&
                                  (d.Wo - 1) * d.gsx + d.gox < d.WoG && d.algo == RAMNET_ALGO_DIRECT);
    if (d.algo == RAMNET_ALGO_WINOGRAD) return launch_wgrad_wino(d, (hipStream_t)stream);
    if (d.algo == RAMNET_ALGO_WINOGRAD_2X4) return launch_wgrad_wino6(d, (hipStream_t)stream);
    if (d.algo == RAMNET_ALGO_DIRECT_SPLIT) {
        RAMNET_CHECK_ARG(gdense);
        return launch_wgrad_dsplit(d, (hipStream_t)stream);
    }
    if (d.algo == RAMNET_ALGO_HEAD) {
        RAMNET_CHECK_ARG(gdense);
        return launch_head_wgrad(d, (hipStream_t)stream);
    }
    RAMNET_CHECK_ARG(d.algo == RAMNET_ALGO_DIRECT && d.in_mode != RAMNET_IN_S2D);

    WgradDerived q;
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d.ntaps; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin, dymax = d.dy[t] > dymax ? d.dy[t] : dymax;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin, dxmax = d.dx[t] > dxmax ? d.dx[t] : dxmax;
    }
    q.dymin = dymin, q.dxmin = dxmin;
    q.gsy = gdense ? 1 : d.gsy, q.gsx = gdense ? 1 : d.gsx, q.goy = d.goy, q.gox = d.gox;
    q.HoG = gdense ? d.Ho : d.HoG, q.WoG = gdense ? d.Wo : d.WoG;
    // two co-resident workgroups per CU hide each other's staging: halve the pixel tile when a full one needs > 80 KB
    q.TH = 8;
    // -> conv_wgrad_kernel<9,2,2>; needs >= 4 pixel tiles per workgroup to amortise its 9 x 64 x 64 atomic epilogue
    const long tiles8 = (long)cdiv(d.Wo, TWID) * cdiv(d.Ho, 8) * d.B;
    const bool wide = d.ntaps <= 9 && d.ntaps > 1 && q.src.Cin >= 64 && d.Cout > 32 &&
                      tiles8 * cdiv(q.src.Cin, 64) * cdiv(d.Cout, 64) >= 2048;
    const int wck = wide ? 64 : WCK;
    if (((7 * d.stride + (dymax - dymin) + 1) * ((TWID - 1) * d.stride + (dxmax - dxmin) + 1) * wck + 128 * 32) * 4 > 80 * 1024) q.TH = 4;
    q.PH = (q.TH - 1) * d.stride + (dymax - dymin) + 1;
    q.PW = (TWID - 1) * d.stride + (dxmax - dxmin) + 1;
    q.tiles_x = cdiv(d.Wo, TWID), q.tiles_y = cdiv(d.Ho, q.TH);
    q.ntiles = q.tiles_x * q.tiles_y * d.B;
    for (int t = 0; t < d.ntaps; ++t) q.toff[t] = ((d.dy[t] - dymin) * q.PW + (d.dx[t] - dxmin)) * wck;
    hipStream_t st = (hipStream_t)stream;
    // Cin < 32: several taps share one 32-row accumulator tile (head convs: 5 or 1 input channels)
    q.tpm = 1;
    if (q.src.Cin <= 16) q.tpm = q.src.Cin <= 4 ? 8 : q.src.Cin <= 8 ? 4 : 2;
    q.ntt = cdiv(d.ntaps, q.tpm);
    // measured (profiles/r01_*layers*): 5x5 layers run 1.5x faster with WBN = 32 (7 tiles/wave, 2 workgroups/CU) than with
    // WBN = 64 (13 tiles/wave, 1 workgroup/CU); 3x3 layers prefer WBN = 64 (5 tiles/wave)
    const bool narrow = d.Cout <= 32 || d.ntaps > 9;
    if (narrow) {                                   // WBN = 32: tap tiles spread over 4 waves
        const int per_wave = cdiv(q.ntt, 4);
        if (per_wave <= 1) return launch_wgrad<1, 1>(d, q, st);
        if (per_wave <= 3) return launch_wgrad<3, 1>(d, q, st);
        if (per_wave <= 4) return launch_wgrad<4, 1>(d, q, st);     // 16 taps: one parity of the folded upsample-conv
        return launch_wgrad<7, 1>(d, q, st);
    }
    if (wide) return launch_wgrad<9, 2, 2>(d, q, st);
    const int per_wave = cdiv(q.ntt, 2);            // WBN = 64: tap tiles spread over 2 wave pairs
    if (per_wave <= 1) return launch_wgrad<1, 2>(d, q, st);
    if (per_wave <= 5) return launch_wgrad<5, 2>(d, q, st);
    return launch_wgrad<13, 2>(d, q, st);
}
